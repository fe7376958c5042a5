// demo-dino -- drop-in of the reference's dinosaur demo (cpp_impl/demos/demo-bundle-adj-dinosaur.cpp:70-243):
// flags --testdata, --f0, --allowed_repr_err (:65-68); reads <testdata>/oxfvisgeom/dinosaur/{dinoPs_as_mat108x4.txt,
// viff.xy}, decomposes the projection matrices, triangulates the tracks and runs ComputeInplace in per-frame-K mode.
// The two data files are not part of the reference tree (testdata/oxfvisgeom/README.md); any directory holding files
// in the same formats works.
#include <cstdio>
#include <string>
#include <vector>

#include "flags.hpp"
#include "srk_ba.h"

int main(int argc, char** argv)
{
    Flags fl;
    fl.Parse(argc, argv);
    const std::string testdata = fl.String("testdata", "NOTFOUND");
    const double f0 = fl.Double("f0", 600), allowed = fl.Double("allowed_repr_err", 1e-5);
    const long max_iterations = fl.Int("max_iterations", 0);
    std::string dir = testdata + "/oxfvisgeom/dinosaur";
    char err[512] = { 0 };
    int64_t N = 0, O = 0;
    int32_t M = 0;
    if (!srk_dino_load(dir.c_str(), f0, &N, &M, &O, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, err, 512)) {
        std::fprintf(stderr, "%s\n", err);
        return 1;
    }
    std::fprintf(stderr, "frames_count=%d\npoints_count=%lld\nf0=%g\n", M, (long long)N, f0);
    std::vector<double> pts(3 * N), R(9 * (size_t)M), T(3 * (size_t)M), K(9 * (size_t)M), uv(2 * O);
    std::vector<int64_t> row_ptr(N + 1);
    std::vector<int32_t> fr(O);
    if (!srk_dino_load(dir.c_str(), f0, &N, &M, &O, pts.data(), R.data(), T.data(), K.data(), row_ptr.data(), fr.data(),
                       uv.data(), err, 512)) {
        std::fprintf(stderr, "%s\n", err);
        return 1;
    }
    srk_ba* h = srk_ba_create(0);
    if (!h) return 2;
    srk_ba_report rep;
    std::fprintf(stderr, "start bundle adjustment..\n");
    int rc = srk_ba_compute_inplace(h, f0, N, pts.data(), M, R.data(), T.data(), K.data(), 0, row_ptr.data(), fr.data(),
                                    uv.data(), allowed > 0 ? &allowed : nullptr, nullptr, max_iterations, &rep);
    if (rc < 0) {
        std::fprintf(stderr, "error: %s\n", srk_ba_last_error(h));
        return 3;
    }
    std::fprintf(stderr, "bundle adjustment finished with result: %d (%s)\n", rc == 0, srk_ba_status_string(rep.status));
    std::printf("{\"result\": %d, \"status\": \"%s\", \"iterations\": %lld, \"attempts\": %lld, \"seen\": %lld, "
                "\"err_initial\": %.17g, \"err_final\": %.17g}\n",
                rc == 0, srk_ba_status_string(rep.status), (long long)rep.iterations, (long long)rep.attempts,
                (long long)rep.seen, rep.err_initial, rep.err_final);
    srk_ba_destroy(h);
    return 0;
}
