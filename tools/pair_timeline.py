"""Timeline of the kernels of the last few LM iterations from a rocprofv3 kernel trace (default bench run: two attempt
slots on two streams): start, duration, stream, so that the overlap of the two attempts can be read off."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
seq = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    seq.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Stream_Id", r.get("Queue_Id", "?"))))
seq.sort()
jac = [i for i, s in enumerate(seq) if s[2].startswith("k_jac_runs")]
i0 = jac[-3]
t0 = seq[i0][0]
last = {}
for s, e, n, q in seq[i0:jac[-2]]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{q:>3} {n[:40]}")
