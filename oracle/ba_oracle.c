/*
 * ba_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See ba_oracle.h.
 *
 * Plain C99 restatement of whigg/surikatoko's Kanatani bundle adjustment.  Every function
 * cites the reference text it follows; citations are relative to
 * /root/reference/cpp_impl/suriko-engine/ ("BA" = src/bundle-adj-kanatani.cpp,
 * "OG" = src/obs-geom.cpp).  Arithmetic is kept in the reference's evaluation order
 * (frames-outer / points-inner accumulation for frame blocks, points-outer for point
 * blocks) so that results agree with the reference's Python prototype to the last bits.
 *
 * Build: gcc -O2 -std=c99 -fPIC -shared (no -ffast-math, no -march: plain IEEE mul/add).
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef ORC_F32 /* the reference's Scalar = float: see "Scalar type" below.  Behind the system headers, in front of everything of this file */
#define double float
#endif
#include "ba_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Baseline variant (iii) of BASELINE.md: the same arithmetic with OpenMP over points / frames / columns.  Off by
 * default (orc_threads == 1: every pragma below is disabled by its if() clause and the loops run in the reference's
 * sequential order, which is what the parity tests pin); bench.py's cpu_baseline leg turns it on for a timing run.
 * The threaded Schur sum and QR update keep every entry's summation order (see there), so results do not change. */
static int orc_threads = 1;
void orc_set_threads(int n) { orc_threads = n < 1 ? 1 : n; }
/* Timing hook of bench.py's cpu_baseline leg ONLY: skip the Householder QR of the reduced system (4/3 n^3 flops: hours
 * at n = 9993) and continue with zero camera corrections, so that the derivative, Schur and back-substitution passes of
 * the 1000-camera scene can be timed on the CPU.  Never set by a test; results are meaningless while it is on. */
/* f32 storage mode of the HIP path (srk_ba_set_storage_precision): the point-frame blocks are rounded to float where the
 * derivative pass stores them; everything computed from them afterwards sees the rounded values.  Off by default. */
static int orc_w_f32 = 0; /* 0: off; 1: the 30 products of a point-frame block rounded to float once; 2: the rank-2 FACTORS rounded */
void orc_set_w_storage_f32(int mode) { orc_w_f32 = mode == 2 ? 2 : (mode != 0); }
static int orc_skip_solve = 0;
void orc_set_skip_solve(int on) { orc_skip_solve = on != 0; }
int orc_get_threads(void) { return orc_threads; }
/* Solver of the two-phase step inside orc_compute_inplace: 0 (default, what every parity test pins) = the reference's
 * Householder QR on dense storage (BA:1911); 1 = BASELINE.md baseline variant (ii), "what a competent CPU port would do":
 * the same Schur arithmetic on skyline storage and a skyline Cholesky (orc_two_phase_skyline).  Used by bench.py's
 * cpu_baseline leg for a complete iteration of the 1000-camera scene (its 9993^2 QR would take hours) and checked
 * against the QR on the small configurations by tests/test_oracle_skyline.py. */
/* Scalar type.  The reference is templated on `Scalar` (rt-config.h:41-48): double by default, float when CMake is given
 * suriko_scalar_type_string=f32 (suriko-engine/CMakeLists.txt:14-15,76-82).  `make libba_oracle_f32.so` compiles THIS file a
 * second time with -DORC_F32 (`double` is then a macro for float behind the system headers): every variable, every array of the interface and every intermediate then is a
 * float, which is what the reference's f32 build computes in (the libm calls go through double and are rounded once, within
 * an ulp of sqrtf and friends).  What changes with the type besides the arithmetic: Eigen's
 * NumTraits<Scalar>::dummy_precision(), the invertibility threshold of computeInverseAndDetWithCheck (:1876), is 1e-12 for
 * double and 1e-5 for float. */
#ifdef ORC_F32
#define ORC_DUMMY_PRECISION 1e-5f
#else
#define ORC_DUMMY_PRECISION 1e-12
#endif
int orc_scalar_bytes(void) { return (int)sizeof(double); } /* 8, or 4 in the f32 build */
static int orc_solver = 0;
void orc_set_solver(int mode) { orc_solver = mode == 1 ? 1 : 0; }
int orc_get_solver(void) { return orc_solver; }

/* (the f32 build below redefines `double`; a float cannot hold a monotonic clock, so seconds are taken relative to the first call) */
static double now_sec(void)
{
    static long long t0_sec = -1;
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    if (t0_sec < 0) t0_sec = (long long)ts.tv_sec;
    return (double)((long long)ts.tv_sec - t0_sec) + (double)(1e-9f * (float)ts.tv_nsec);
}

/* ---------------------------------------------------------------- small helpers */

/* include/suriko/approx-alg.h:8-16: |a-b| <= atol + rtol*|max(a,b)|  (sic: max before abs) */
int orc_is_close(double a, double b, double rtol, double atol)
{
    double mx = a > b ? a : b;
    return fabs(a - b) <= (atol + rtol * fabs(mx));
}
#define ISCLOSE_DEF(a, b) orc_is_close((a), (b), 1.0e-5, 1.0e-8)

static void mat3_mul(const double A[9], const double B[9], double C[9])
{
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            t[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
    memcpy(C, t, sizeof t);
}
static void mat3_tr(const double A[9], double At[9])
{
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[i * 3 + j] = A[j * 3 + i];
    memcpy(At, t, sizeof t);
}
static void mat3_vec(const double A[9], const double x[3], double y[3])
{
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = A[i * 3 + 0] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
    y[0] = t[0]; y[1] = t[1]; y[2] = t[2];
}
static double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* OG:506-512 SkewSymmetricMat */
void orc_skew(const double v[3], double S[9])
{
    S[0] = 0;     S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2];  S[4] = 0;     S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0];  S[8] = 0;
}

/* OG:117-122 SE3Inv */
void orc_se3_inv(const double R[9], const double T[3], double Ri[9], double Ti[3])
{
    double Rt[9], t[3];
    mat3_tr(R, Rt);
    mat3_vec(Rt, T, t);
    memcpy(Ri, Rt, sizeof Rt);
    Ti[0] = -t[0]; Ti[1] = -t[1]; Ti[2] = -t[2];
}
/* OG:131-139 SE3Apply */
static void se3_apply(const double R[9], const double T[3], const double x[3], double y[3])
{
    double t[3];
    mat3_vec(R, x, t);
    y[0] = t[0] + T[0]; y[1] = t[1] + T[1]; y[2] = t[2] + T[2];
}

/* OG:520-551 RotMatFromUnityDirAndAngle (Rodrigues: I + s*K + (1-c)*K*K) */
int orc_rot_from_unity_dir_and_angle(const double dir[3], double ang, double R[9], int check_input)
{
    if (check_input) {
        double len = norm3(dir);
        if (!ISCLOSE_DEF(1.0, len)) return 0;
        if (ISCLOSE_DEF(0.0, ang)) return 0;
    }
    double s = sin(ang), c = cos(ang);
    double K[9], KK[9];
    orc_skew(dir, K);
    mat3_mul(K, K, KK);
    for (int i = 0; i < 9; ++i) {
        double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
        R[i] = id + s * K[i] + (1 - c) * KK[i];
    }
    return 1;
}

/* OG:553-561 RotMatFromAxisAngle */
int orc_rot_from_axis_angle(const double w[3], double R[9])
{
    double ang = norm3(w);
    if (ISCLOSE_DEF(0.0, ang)) return 0;
    double dir[3] = { w[0] / ang, w[1] / ang, w[2] / ang };
    return orc_rot_from_unity_dir_and_angle(dir, ang, R, 0);
}

/* OG:563-593 LogSO3 (float literals kept: 0.5f, 1.0f, 1e-3f) */
int orc_log_so3(const double R[9], double dir[3], double* ang)
{
    double cos_ang = 0.5 * (R[0] + R[4] + R[8] - 1);
    if (cos_ang < -1) cos_ang = -1;
    if (cos_ang > 1) cos_ang = 1;
    double sin_ang = sqrt(1.0 - cos_ang * cos_ang);
    double atol = (double)1e-3f;
    if (orc_is_close(0.0, sin_ang, 0.0, atol)) return 0;
    dir[0] = R[2 * 3 + 1] - R[1 * 3 + 2];
    dir[1] = R[0 * 3 + 2] - R[2 * 3 + 0];
    dir[2] = R[1 * 3 + 0] - R[0 * 3 + 1];
    double k = 0.5 / sin_ang;
    dir[0] *= k; dir[1] *= k; dir[2] *= k;
    double len = norm3(dir);
    double il = 1 / len;
    dir[0] *= il; dir[1] *= il; dir[2] *= il;
    *ang = acos(cos_ang);
    return 1;
}
/* OG:595-604 AxisAngleFromRotMat */
int orc_axis_angle_from_rot(const double R[9], double w[3])
{
    double dir[3], ang;
    if (!orc_log_so3(R, dir, &ang)) return 0;
    w[0] = dir[0] * ang; w[1] = dir[1] * ang; w[2] = dir[2] * ang;
    return 1;
}

/* src/virt-world/scene-generator.cpp:9-55 GenerateCircleCameraShots */
void orc_circle_camera_shots(const double center[3], double radius, double ascent_z,
                             int32_t n, const double* angles, double* cam_R, double* cam_T)
{
    for (int32_t a = 0; a < n; ++a) {
        double ang = angles[a];
        double c2c[3] = { radius * cos(ang), radius * sin(ang), ascent_z }; /* center_to_cam_pos */
        double shift[3] = { center[0] + c2c[0], center[1] + c2c[1], center[2] + c2c[2] };
        /* cam_from_world = [I | -shift] */
        double R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        double T[3] = { -shift[0], -shift[1], -shift[2] };
        /* rotate OY around OZ towards the centre */
        double d[3] = { -shift[0], -shift[1], 0 };
        double dl = norm3(d);
        d[0] /= dl; d[1] /= dl; d[2] /= dl;
        double oy[3] = { 0, 1, 0 }, oz[3] = { 0, 0, 1 };
        double yaw = acos(oy[0] * d[0] + oy[1] * d[1] + oy[2] * d[2]);
        double cr[3];
        cross3(oy, d, cr);
        double dotz = cr[0] * oz[0] + cr[1] * oz[1] + cr[2] * oz[2];
        int sgn = dotz >= 0 ? 1 : -1; /* approx-alg.h:41 Sign */
        yaw *= sgn;
        double Rz[9];
        if (!orc_rot_from_unity_dir_and_angle(oz, -yaw, Rz, 1)) { /* OG:1060-1066 RotMat: identity on failure */
            double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
            memcpy(Rz, I, sizeof I);
        }
        mat3_mul(Rz, R, R);
        mat3_vec(Rz, T, T);
        double look_down = atan2(c2c[2], sqrt(c2c[0] * c2c[0] + c2c[1] * c2c[1]));
        double ox[3] = { 1, 0, 0 };
        double Rx[9];
        if (!orc_rot_from_unity_dir_and_angle(ox, look_down + M_PI / 2, Rx, 1)) {
            double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
            memcpy(Rx, I, sizeof I);
        }
        mat3_mul(Rx, R, R);
        mat3_vec(Rx, T, T);
        memcpy(cam_R + 9 * a, R, sizeof R);
        memcpy(cam_T + 3 * a, T, sizeof T);
    }
}

/* include/suriko/eigen-helpers.hpp:10-83 RemoveRowsAndColsInplace (row-major int64 payload) */
void orc_remove_rows_cols(int64_t rows, int64_t cols, int64_t* mat,
                          int32_t n_rm_rows, const int64_t* rm_rows,
                          int32_t n_rm_cols, const int64_t* rm_cols,
                          int64_t* new_rows_out, int64_t* new_cols_out)
{
    int64_t new_rows = rows - n_rm_rows, new_cols = cols - n_rm_cols;
    int64_t* tmp = (int64_t*)malloc(sizeof(int64_t) * (size_t)(rows * cols + 1));
    int64_t rd = 0;
    int32_t ri = 0;
    for (int64_t r = 0; r < rows; ++r) {
        if (ri < n_rm_rows && rm_rows[ri] == r) { ++ri; continue; }
        int64_t cd = 0;
        int32_t ci = 0;
        for (int64_t c = 0; c < cols; ++c) {
            if (ci < n_rm_cols && rm_cols[ci] == c) { ++ci; continue; }
            tmp[rd * new_cols + cd] = mat[r * cols + c];
            ++cd;
        }
        ++rd;
    }
    if (new_rows == 0) new_cols = 0; /* :75-78 an empty dimension empties the matrix */
    if (new_cols == 0) new_rows = 0;
    memcpy(mat, tmp, sizeof(int64_t) * (size_t)(new_rows * new_cols));
    free(tmp);
    *new_rows_out = new_rows;
    *new_cols_out = new_cols;
}

/* Eigen Matrix3::computeInverseAndDetWithCheck (cofactor inverse; invertible iff |det| > 1e-12,
 * Eigen's NumTraits<double>::dummy_precision()).  Call sites BA:1876,1936. */
int orc_inverse3x3_with_check(const double A[9], double Ainv[9], double* det_out)
{
#define A_(i, j) A[(i) * 3 + (j)]
    double c00 = A_(1, 1) * A_(2, 2) - A_(1, 2) * A_(2, 1);
    double c10 = A_(1, 2) * A_(2, 0) - A_(1, 0) * A_(2, 2); /* cofactor(0,1) */
    double c20 = A_(1, 0) * A_(2, 1) - A_(1, 1) * A_(2, 0); /* cofactor(0,2) */
    double det = A_(0, 0) * c00 + A_(0, 1) * c10 + A_(0, 2) * c20;
    if (det_out) *det_out = det;
    if (!(fabs(det) > ORC_DUMMY_PRECISION)) return 0;
    double id = 1 / det;
    Ainv[0] = c00 * id;
    Ainv[1] = (A_(0, 2) * A_(2, 1) - A_(0, 1) * A_(2, 2)) * id;
    Ainv[2] = (A_(0, 1) * A_(1, 2) - A_(0, 2) * A_(1, 1)) * id;
    Ainv[3] = c10 * id;
    Ainv[4] = (A_(0, 0) * A_(2, 2) - A_(0, 2) * A_(2, 0)) * id;
    Ainv[5] = (A_(0, 2) * A_(1, 0) - A_(0, 0) * A_(1, 2)) * id;
    Ainv[6] = c20 * id;
    Ainv[7] = (A_(0, 1) * A_(2, 0) - A_(0, 0) * A_(2, 1)) * id;
    Ainv[8] = (A_(0, 0) * A_(1, 1) - A_(0, 1) * A_(1, 0)) * id;
#undef A_
    return 1;
}

/* Eigen HouseholderQR (unblocked: Householder.h makeHouseholder / applyHouseholderOnTheLeft,
 * HouseholderQR.h householder_qr_inplace_unblocked + _solve_impl).  Call site BA:1911.
 * A column-major n x n, overwritten.  Returns 1 iff the solution is all finite. */
int orc_householder_qr_solve(int64_t n, double* A, const double* b, double* x)
{
    double* tau = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double* c = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    memcpy(c, b, sizeof(double) * (size_t)n);
    const double tiny = 2.2250738585072014e-308;
    for (int64_t k = 0; k < n; ++k) {
        double* col = A + k * n + k; /* tail of column k, length n-k */
        int64_t m = n - k;
        double c0 = col[0];
        double tail_sq = 0;
        for (int64_t i = 1; i < m; ++i) tail_sq += col[i] * col[i];
        double beta, t;
        if (tail_sq <= tiny) {
            t = 0;
            beta = c0;
            for (int64_t i = 1; i < m; ++i) col[i] = 0;
        } else {
            beta = sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0) beta = -beta;
            double inv = c0 - beta;
            for (int64_t i = 1; i < m; ++i) col[i] = col[i] / inv;
            t = (beta - c0) / beta;
        }
        tau[k] = t;
        col[0] = beta;
        /* apply H_k to the trailing columns and to c */
        if (m == 1) {
            /* rows()==1: *this *= (1 - tau) -- no trailing columns when k == n-1 */
            c[k] *= (1 - t);
        } else if (t != 0) {
#pragma omp parallel for schedule(static) num_threads(orc_threads) if (orc_threads > 1 && (n - k) * m > 400000)
            for (int64_t j = k + 1; j < n; ++j) {
                double* cj = A + j * n + k;
                double tmp = 0;
                for (int64_t i = 1; i < m; ++i) tmp += col[i] * cj[i];
                tmp += cj[0];
                cj[0] -= t * tmp;
                double tt = t * tmp;
                for (int64_t i = 1; i < m; ++i) cj[i] -= col[i] * tt;
            }
            double tmp = 0;
            for (int64_t i = 1; i < m; ++i) tmp += col[i] * c[k + i];
            tmp += c[k];
            c[k] -= t * tmp;
            double tt = t * tmp;
            for (int64_t i = 1; i < m; ++i) c[k + i] -= col[i] * tt;
        }
    }
    /* back substitution with R (upper triangle of A) */
    for (int64_t i = n - 1; i >= 0; --i) {
        double s = c[i];
        for (int64_t j = i + 1; j < n; ++j) s -= A[j * n + i] * x[j];
        x[i] = s / A[i * n + i];
    }
    int finite = 1;
    for (int64_t i = 0; i < n; ++i)
        if (!isfinite(x[i])) finite = 0;
    free(tau);
    free(c);
    return finite;
}

/* ---------------------------------------------------------------- normalisation */

/* BA:143-162 NormalizeRT */
static void normalize_rt(const double Rk[9], const double Tk[3], const double R0[9], const double T0[3], double s,
                         double Rn[9], double Tn[3])
{
    double R0t[9], RR[9], v[3];
    mat3_tr(R0, R0t);
    mat3_mul(Rk, R0t, RR);
    mat3_vec(RR, T0, v);
    memcpy(Rn, RR, sizeof RR);
    for (int i = 0; i < 3; ++i) Tn[i] = (Tk[i] - v[i]) * s;
}
/* BA:164-177 RevertRT */
static void revert_rt(const double Rk[9], const double Tk[3], const double R0[9], const double T0[3], double s,
                      double Rn[9], double Tn[3])
{
    double RR[9], v[3];
    mat3_mul(Rk, R0, RR);
    mat3_vec(Rk, T0, v);
    for (int i = 0; i < 3; ++i) Tn[i] = Tk[i] / s + v[i];
    memcpy(Rn, RR, sizeof RR);
}

/* BA:288-333 CheckWorldIsNormalized */
int orc_check_world_is_normalized(int32_t n_frames, const double* cam_R, const double* cam_T, double t1y,
                                  int32_t comp)
{
    if (n_frames < 2) return 0;
    const double atol = 1e-3;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double e = r == c ? 1.0 : 0.0;
            if (!orc_is_close(e, cam_R[r * 3 + c], atol, atol)) return 0;
        }
    if (norm3(cam_T) >= atol) return 0;
    double Ri[9], Ti[3];
    orc_se3_inv(cam_R + 9, cam_T + 3, Ri, Ti);
    if (!orc_is_close(t1y, fabs(Ti[comp]), atol, 1.0e-8)) return 0;
    return 1;
}

/* BA:203-247 NormalizeWorldInplaceInternal (+ :277-286) */
int orc_normalize_scene(int64_t n_points, double* points, int32_t n_frames, double* cam_R, double* cam_T,
                        double t1y, int32_t comp, orc_normalizer* out)
{
    if (n_frames < 2 || comp < 0 || comp > 2) return 0;
    /* cam0_from1 = SE3AFromB(cam0, cam1) = Compose(cam0, Inv(cam1))  (OG:141-150) */
    double R1i[9], T1i[3], T01[3], v[3];
    orc_se3_inv(cam_R + 9, cam_T + 3, R1i, T1i);
    mat3_vec(cam_R, T1i, v);
    for (int i = 0; i < 3; ++i) T01[i] = v[i] + cam_T[i];
    double shift = T01[comp];
    /* BA:215-217: IsClose(0, shift, atol) -- the third argument is rtol */
    if (orc_is_close(0.0, shift, 1e-5, 1.0e-8)) return 0;
    double s = t1y / fabs(shift);
    memcpy(out->R0, cam_R, sizeof out->R0);
    memcpy(out->T0, cam_T, sizeof out->T0);
    out->world_scale = s;
    for (int32_t j = 0; j < n_frames; ++j) {
        double Rn[9], Tn[3];
        normalize_rt(cam_R + 9 * j, cam_T + 3 * j, out->R0, out->T0, s, Rn, Tn);
        memcpy(cam_R + 9 * j, Rn, sizeof Rn);
        memcpy(cam_T + 3 * j, Tn, sizeof Tn);
    }
    for (int64_t i = 0; i < n_points; ++i) { /* BA:179-199 NormalizeOrRevertPoint */
        double y[3];
        se3_apply(out->R0, out->T0, points + 3 * i, y);
        points[3 * i + 0] = y[0] * s;
        points[3 * i + 1] = y[1] * s;
        points[3 * i + 2] = y[2] * s;
    }
    return 1;
}

/* BA:249-270 RevertNormalization */
void orc_revert_normalization(int64_t n_points, double* points, int32_t n_frames, double* cam_R, double* cam_T,
                              const orc_normalizer* nrm)
{
    double s = nrm->world_scale;
    double R0t[9];
    mat3_tr(nrm->R0, R0t);
    for (int64_t i = 0; i < n_points; ++i) {
        double is = 1 / s;
        double t[3] = { points[3 * i] * is - nrm->T0[0], points[3 * i + 1] * is - nrm->T0[1],
                        points[3 * i + 2] * is - nrm->T0[2] };
        mat3_vec(R0t, t, points + 3 * i);
    }
    for (int32_t j = 0; j < n_frames; ++j) {
        double Rn[9], Tn[3];
        revert_rt(cam_R + 9 * j, cam_T + 3 * j, nrm->R0, nrm->T0, s, Rn, Tn);
        memcpy(cam_R + 9 * j, Rn, sizeof Rn);
        memcpy(cam_T + 3 * j, Tn, sizeof Tn);
    }
}

/* ---------------------------------------------------------------- frame-major index */

typedef struct {
    int64_t* col_ptr; /* [M+1] */
    int64_t* obs;     /* [O] observation index, ordered by (frame, pnt_ind) */
    int64_t* pnt;     /* [O] pnt_ind of that observation */
} csc_t;

static void csc_build(int64_t N, int32_t M, const int64_t* row_ptr, const int32_t* obs_frame, csc_t* c)
{
    int64_t O = row_ptr[N];
    c->col_ptr = (int64_t*)calloc((size_t)M + 2, sizeof(int64_t));
    c->obs = (int64_t*)malloc(sizeof(int64_t) * (size_t)(O > 0 ? O : 1));
    c->pnt = (int64_t*)malloc(sizeof(int64_t) * (size_t)(O > 0 ? O : 1));
    for (int64_t o = 0; o < O; ++o) c->col_ptr[obs_frame[o] + 1]++;
    for (int32_t j = 0; j < M; ++j) c->col_ptr[j + 1] += c->col_ptr[j];
    int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * ((size_t)M + 1));
    memcpy(fill, c->col_ptr, sizeof(int64_t) * ((size_t)M + 1));
    for (int64_t i = 0; i < N; ++i)
        for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o) {
            int64_t k = fill[obs_frame[o]]++;
            c->obs[k] = o;
            c->pnt[k] = i;
        }
    free(fill);
}
static void csc_free(csc_t* c)
{
    free(c->col_ptr);
    free(c->obs);
    free(c->pnt);
}

/* ---------------------------------------------------------------- reprojection error */

/* BA:410-490 ReprojErrorWithOverlap (no patches): frames outer, tracks inner */
double orc_reproj_error(double f0, int64_t N, const double* points, int32_t M, const double* cam_R,
                        const double* cam_T, const double* K, int32_t shared_k, const int64_t* row_ptr,
                        const int32_t* obs_frame, const double* obs_uv, int64_t* seen)
{
    csc_t c;
    csc_build(N, M, row_ptr, obs_frame, &c);
    double err_sum = 0;
    int64_t cnt = 0;
    for (int32_t j = 0; j < M; ++j) {
        const double* R = cam_R + 9 * j;
        const double* T = cam_T + 3 * j;
        const double* Kj = shared_k ? K : K + 9 * j;
        for (int64_t k = c.col_ptr[j]; k < c.col_ptr[j + 1]; ++k) {
            int64_t o = c.obs[k];
            const double* X = points + 3 * c.pnt[k];
            double xc[3], xi[3];
            se3_apply(R, T, X, xc);
            mat3_vec(Kj, xc, xi);
            double x = xi[0] / xi[2];
            double y = xi[1] / xi[2];
            double dx = x - obs_uv[2 * o] / f0;
            double dy = y - obs_uv[2 * o + 1] / f0;
            err_sum += dx * dx + dy * dy;
            cnt += 1;
        }
    }
    csc_free(&c);
    if (seen) *seen = cnt;
    return err_sum;
}

/* multi-view-factorization.cpp:415-475 MultiViewIterativeFactorizer::ReprojError: the scorer the MVF driver calls
 * before deciding to run BA (:372-379) -- frames outer, tracks inner, observations whose homogeneous image point
 * has |z| <= 1e-5 (IsCloseAbs, approx-alg.h:19-23) are skipped; returns 0 (false) when nothing was summed. */
int orc_reproj_error_mvf(double f0, int64_t N, const double* points, int32_t M, const double* cam_R,
                         const double* cam_T, const double* K, int32_t shared_k, const int64_t* row_ptr,
                         const int32_t* obs_frame, const double* obs_uv, double* reproj_err, int64_t* summands)
{
    csc_t c;
    csc_build(N, M, row_ptr, obs_frame, &c);
    double err_sum = 0;
    int64_t cnt = 0;
    for (int32_t j = 0; j < M; ++j) {
        const double* R = cam_R + 9 * j;
        const double* T = cam_T + 3 * j;
        const double* Kj = shared_k ? K : K + 9 * j;
        for (int64_t k = c.col_ptr[j]; k < c.col_ptr[j + 1]; ++k) {
            int64_t o = c.obs[k];
            const double* X = points + 3 * c.pnt[k];
            double xc[3], xi[3];
            se3_apply(R, T, X, xc);
            mat3_vec(Kj, xc, xi);
            if (fabs(0 - xi[2]) <= 1e-5) continue; /* :455-457 */
            double x = xi[0] / xi[2];
            double y = xi[1] / xi[2];
            double dx = x - obs_uv[2 * o] / f0;
            double dy = y - obs_uv[2 * o + 1] / f0;
            err_sum += dx * dx + dy * dy;
            cnt += 1;
        }
    }
    csc_free(&c);
    if (summands) *summands = cnt;
    if (cnt == 0) return 0; /* :470-471 */
    *reproj_err = err_sum;
    return 1;
}

/* ---------------------------------------------------------------- multi-view-factorization steps
 * The reference computes these with Eigen::JacobiSVD (an un-vendored dependency); the restatement uses a one-sided
 * (Hestenes) Jacobi SVD, which yields the same singular vectors up to sign -- and ProjectOntoSO3 is sign-invariant. */

/* one-sided Jacobi SVD of A (m x n, row-major, m >= n): A is overwritten by U*diag(s) column-wise, V (n x n row-major)
 * accumulates the rotations; singular values = column norms (unsorted). */
static void svd_hestenes(int64_t m, int n, double* A, double* V, double* sv)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double a = 0, b = 0, c = 0;
                for (int64_t i = 0; i < m; ++i) {
                    double x = A[i * n + p], y = A[i * n + q];
                    a += x * x; b += y * y; c += x * y;
                }
                if (c == 0) continue;
                double lim = sqrt(a * b);
                if (fabs(c) <= 1e-300 + 1e-17 * lim) continue;
                if (fabs(c) / (lim > 0 ? lim : 1) > off) off = fabs(c) / (lim > 0 ? lim : 1);
                double zeta = (b - a) / (2 * c);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                double cs = 1 / sqrt(1 + t * t), sn = cs * t;
                for (int64_t i = 0; i < m; ++i) {
                    double x = A[i * n + p], y = A[i * n + q];
                    A[i * n + p] = cs * x - sn * y;
                    A[i * n + q] = sn * x + cs * y;
                }
                for (int i = 0; i < n; ++i) {
                    double x = V[i * n + p], y = V[i * n + q];
                    V[i * n + p] = cs * x - sn * y;
                    V[i * n + q] = sn * x + cs * y;
                }
            }
        if (off < 1e-15) break;
    }
    for (int j = 0; j < n; ++j) {
        double a = 0;
        for (int64_t i = 0; i < m; ++i) a += A[i * n + j] * A[i * n + j];
        sv[j] = sqrt(a);
    }
}

/* multi-view-factorization.cpp:79-104 ProjectOntoSO3 (MASKS 8.41, 8.42).  R row-major.  returns 0 when det S ~ 0 */
int orc_project_onto_so3(const double R_noisy[9], const double T_noisy[3], double R_out[9], double T_out[3])
{
    double A[9], V[9], sv[3], U[9];
    memcpy(A, R_noisy, sizeof A);
    svd_hestenes(3, 3, A, V, sv);
    double det_S = sv[0] * sv[1] * sv[2];
    if (ISCLOSE_DEF(0, det_S)) return 0; /* :88-89 */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) U[3 * i + j] = A[3 * i + j] / sv[j];
    double ng[9]; /* U V^T */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) ng[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
    double det = ng[0] * (ng[4] * ng[8] - ng[5] * ng[7]) - ng[1] * (ng[3] * ng[8] - ng[5] * ng[6]) +
                 ng[2] * (ng[3] * ng[7] - ng[4] * ng[6]);
    int sign = det >= 0 ? 1 : -1; /* approx-alg.h:41 */
    for (int i = 0; i < 9; ++i) R_out[i] = sign * ng[i];
    double s = sign / cbrt(det_S);
    for (int i = 0; i < 3; ++i) T_out[i] = s * T_noisy[i];
    return 1;
}

/* multi-view-factorization.cpp:107-189 FindRelativeMotionMultiPoints: rows [x2]x (kron(x1^T, I) | alpha I) of the
 * 3P x 12 system, r_and_t = right singular vector of the smallest singular value (vec(R) column-major, then T),
 * then ProjectOntoSO3.  x_anchor / x_target: homogeneous image coordinates [P][3]. */
int orc_mvf_relative_motion(int64_t P, const double* x_anchor, const double* x_target, const double* depth_anchor,
                            double R_out[9], double T_out[3])
{
    if (P < 6) return 0; /* 2 independent equations per point, 11 needed */
    double* A = (double*)calloc((size_t)(3 * P * 12), sizeof(double));
    for (int64_t i = 0; i < P; ++i) {
        const double* c1 = x_anchor + 3 * i;
        const double* c2 = x_target + 3 * i;
        double sk[9] = { 0, -c2[2], c2[1], c2[2], 0, -c2[0], -c2[1], c2[0], 0 }; /* obs-geom.cpp:512-518 */
        double alpha = 1 / depth_anchor[i];
        for (int r = 0; r < 3; ++r) {
            double* row = A + (3 * i + r) * 12;
            for (int comp = 0; comp < 3; ++comp)
                for (int cc = 0; cc < 3; ++cc) row[3 * comp + cc] = c1[comp] * sk[3 * r + cc];
            for (int cc = 0; cc < 3; ++cc) row[9 + cc] = alpha * sk[3 * r + cc];
        }
    }
    double V[144], sv[12];
    svd_hestenes(3 * P, 12, A, V, sv);
    free(A);
    int jmin = 0;
    for (int j = 1; j < 12; ++j)
        if (sv[j] < sv[jmin]) jmin = j;
    double Rn[9], Tn[3];
    for (int col = 0; col < 3; ++col) /* column-major vec(R) (:174) */
        for (int row = 0; row < 3; ++row) Rn[3 * row + col] = V[(3 * col + row) * 12 + jmin];
    for (int i = 0; i < 3; ++i) Tn[i] = V[(9 + i) * 12 + jmin];
    return orc_project_onto_so3(Rn, Tn, R_out, T_out);
}

/* multi-view-factorization.cpp:223-253 Estimate3DPointDepthFromFrames (MASKS 8.44) for ONE track: observation 0 is
 * the base frame; frame_from_base = SE3AFromB(frame_i_from_world, base_from_world) (:205-213, obs-geom.cpp:147-150). */
double orc_mvf_point_depth(int64_t n_obs, const int32_t* frame, const double* x_meter, const double* cam_R,
                           const double* cam_T)
{
    const double* Rb = cam_R + 9 * frame[0];
    const double* Tb = cam_T + 3 * frame[0];
    const double* x1 = x_meter;
    double num = 0, den = 0;
    for (int64_t i = 1; i < n_obs; ++i) {
        const double* Ri = cam_R + 9 * frame[i];
        const double* Ti = cam_T + 3 * frame[i];
        const double* xi = x_meter + 3 * i;
        double Rr[9], Tr[3], v[3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Rr[3 * a + b] = Ri[3 * a] * Rb[3 * b] + Ri[3 * a + 1] * Rb[3 * b + 1] + Ri[3 * a + 2] * Rb[3 * b + 2];
        for (int a = 0; a < 3; ++a) Tr[a] = Ti[a] - (Rr[3 * a] * Tb[0] + Rr[3 * a + 1] * Tb[1] + Rr[3 * a + 2] * Tb[2]);
        for (int a = 0; a < 3; ++a) v[a] = Rr[3 * a] * x1[0] + Rr[3 * a + 1] * x1[1] + Rr[3 * a + 2] * x1[2];
        double h1[3] = { xi[1] * Tr[2] - xi[2] * Tr[1], xi[2] * Tr[0] - xi[0] * Tr[2], xi[0] * Tr[1] - xi[1] * Tr[0] };
        double h2[3] = { xi[1] * v[2] - xi[2] * v[1], xi[2] * v[0] - xi[0] * v[2], xi[0] * v[1] - xi[1] * v[0] };
        num += h1[0] * h2[0] + h1[1] * h2[1] + h1[2] * h2[2];
        den += h1[0] * h1[0] + h1[1] * h1[1] + h1[2] * h1[2];
    }
    double alpha = -num / den;
    return 1 / alpha;
}

/* ---------------------------------------------------------------- derivatives */

/* BA:1528-1537 FirstDerivFromPqrDerivative (formula 8) */
static double first_deriv(double f0, const double pqr[3], const double uv[2], double gp, double gq, double gr)
{
    double result = (pqr[0] / pqr[2] - uv[0] / f0) * (pqr[2] * gp - pqr[0] * gr) +
                    (pqr[1] / pqr[2] - uv[1] / f0) * (pqr[2] * gq - pqr[1] * gr);
    result *= 2 / (pqr[2] * pqr[2]);
    return result;
}
/* BA:1540-1549 SecondDerivFromPqrDerivative (formula 9) */
static double second_deriv(const double pqr[3], double gp1, double gq1, double gr1, double gp2, double gq2,
                           double gr2)
{
    double s = (pqr[2] * gp1 - pqr[0] * gr1) * (pqr[2] * gp2 - pqr[0] * gr2) +
               (pqr[2] * gq1 - pqr[1] * gr1) * (pqr[2] * gq2 - pqr[1] * gr2);
    s *= 2 / (pqr[2] * pqr[2] * pqr[2] * pqr[2]);
    return s;
}

/* BA:1450-1455 ComputePointPqrDerivatives: row v = d(p,q,r)/d(X,Y,Z)[v] = column v of P = K[R|T] */
static void point_pqr_derivs(const double K[9], const double R[9], double d[3][3])
{
    double KR[9];
    mat3_mul(K, R, KR);
    for (int v = 0; v < 3; ++v)
        for (int c = 0; c < 3; ++c) d[v][c] = KR[c * 3 + v];
}

/* BA:1457-1525 ComputeFramePqrDerivatives: d[var][p|q|r], var order [fx fy u0 v0 Tx Ty Tz Wx Wy Wz] */
static void frame_pqr_derivs(double f0, const double K[9], const double R[9], const double T[3], const double X[3],
                             double d[10][3])
{
    double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    double xc[3], pqr[3];
    se3_apply(R, T, X, xc);
    mat3_vec(K, xc, pqr);
    d[0][0] = (1 / fx) * pqr[0] - u0 / (f0 * fx) * pqr[2]; d[0][1] = 0; d[0][2] = 0;
    d[1][0] = 0; d[1][1] = (1 / fy) * pqr[1] - v0 / (f0 * fy) * pqr[2]; d[1][2] = 0;
    d[2][0] = (1 / f0) * pqr[2]; d[2][1] = 0; d[2][2] = 0;
    d[3][0] = 0; d[3][1] = (1 / f0) * pqr[2]; d[3][2] = 0;
    double Rd[9], Td[3];
    orc_se3_inv(R, T, Rd, Td);
    double rot1[3], rot2[3], rot3[3];
    for (int c = 0; c < 3; ++c) {
        rot1[c] = fx * Rd[c * 3 + 0] + u0 * Rd[c * 3 + 2];
        rot2[c] = fy * Rd[c * 3 + 1] + v0 * Rd[c * 3 + 2];
        rot3[c] = f0 * Rd[c * 3 + 2];
        d[4 + c][0] = -rot1[c];
        d[4 + c][1] = -rot2[c];
        d[4 + c][2] = -rot3[c];
    }
    double t[3] = { X[0] - Td[0], X[1] - Td[1], X[2] - Td[2] };
    double c1[3], c2[3], c3[3];
    cross3(rot1, t, c1);
    cross3(rot2, t, c2);
    cross3(rot3, t, c3);
    for (int c = 0; c < 3; ++c) {
        d[7 + c][0] = c1[c];
        d[7 + c][1] = c2[c];
        d[7 + c][2] = c3[c];
    }
}

/* BA:1140-1448 ComputeCloseFormReprErrorDerivatives, block-sparse storage */
void orc_derivatives(double f0, int64_t N, const double* points, int32_t M, const double* cam_R,
                     const double* cam_T, const double* K, int32_t shared_k, const int64_t* row_ptr,
                     const int32_t* obs_frame, const double* obs_uv, double* gradE, double* Vpp, double* Uff,
                     double* Wpf)
{
    int64_t O = row_ptr[N];
    memset(gradE, 0, sizeof(double) * (size_t)(3 * N + 10 * (int64_t)M));
    memset(Vpp, 0, sizeof(double) * (size_t)(9 * N));
    memset(Uff, 0, sizeof(double) * (size_t)(100 * (int64_t)M));
    memset(Wpf, 0, sizeof(double) * (size_t)(30 * O));

    /* points loop BA:1163-1221 */
#pragma omp parallel for schedule(static) num_threads(orc_threads) if (orc_threads > 1)
    for (int64_t i = 0; i < N; ++i) {
        const double* X = points + 3 * i;
        double* gp = gradE + 3 * i;
        for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o) {
            int32_t j = obs_frame[o];
            const double* R = cam_R + 9 * j;
            const double* T = cam_T + 3 * j;
            const double* Kj = shared_k ? K : K + 9 * j;
            double xc[3], pqr[3], pd[3][3];
            se3_apply(R, T, X, xc);
            mat3_vec(Kj, xc, pqr);
            point_pqr_derivs(Kj, R, pd);
            for (int v = 0; v < 3; ++v) gp[v] += first_deriv(f0, pqr, obs_uv + 2 * o, pd[v][0], pd[v][1], pd[v][2]);
            for (int v1 = 0; v1 < 3; ++v1)
                for (int v2 = 0; v2 < 3; ++v2)
                    Vpp[9 * i + 3 * v1 + v2] +=
                        second_deriv(pqr, pd[v1][0], pd[v1][1], pd[v1][2], pd[v2][0], pd[v2][1], pd[v2][2]);
        }
    }

    /* frames loop BA:1270-1359 */
    csc_t c;
    csc_build(N, M, row_ptr, obs_frame, &c);
#pragma omp parallel for schedule(dynamic, 1) num_threads(orc_threads) if (orc_threads > 1)
    for (int32_t j = 0; j < M; ++j) {
        const double* R = cam_R + 9 * j;
        const double* T = cam_T + 3 * j;
        const double* Kj = shared_k ? K : K + 9 * j;
        double* gf = gradE + 3 * N + 10 * (int64_t)j;
        double* U = Uff + 100 * (int64_t)j;
        for (int64_t k = c.col_ptr[j]; k < c.col_ptr[j + 1]; ++k) {
            int64_t o = c.obs[k];
            const double* X = points + 3 * c.pnt[k];
            double xc[3], pqr[3], fd[10][3], pd[3][3];
            se3_apply(R, T, X, xc);
            mat3_vec(Kj, xc, pqr);
            frame_pqr_derivs(f0, Kj, R, T, X, fd);
            for (int v = 0; v < 10; ++v) gf[v] += first_deriv(f0, pqr, obs_uv + 2 * o, fd[v][0], fd[v][1], fd[v][2]);
            for (int v1 = 0; v1 < 10; ++v1)
                for (int v2 = 0; v2 < 10; ++v2)
                    U[10 * v1 + v2] +=
                        second_deriv(pqr, fd[v1][0], fd[v1][1], fd[v1][2], fd[v2][0], fd[v2][1], fd[v2][2]);
            point_pqr_derivs(Kj, R, pd);
            double* W = Wpf + 30 * o;
            if (orc_w_f32 == 2) {
                /* what the HIP path's f32 storage mode keeps: formula 9 is W[pv][fv] = Ap[pv] Af[fv] + Bp[pv] Bf[fv] with
                 * A(v) = (r p' - p r') sqrt(2) / r^2, B(v) = (r q' - q r') sqrt(2) / r^2 (BA:1540-1549 regrouped); the
                 * library stores those FACTORS as floats and forms the products in double */
                const double sc = 1.4142135623730951 / (pqr[2] * pqr[2]);
                float Apf[3], Bpf[3], Aff[10], Bff[10];
                for (int pv = 0; pv < 3; ++pv) {
                    Apf[pv] = (float)((pqr[2] * pd[pv][0] - pqr[0] * pd[pv][2]) * sc);
                    Bpf[pv] = (float)((pqr[2] * pd[pv][1] - pqr[1] * pd[pv][2]) * sc);
                }
                for (int fv = 0; fv < 10; ++fv) {
                    Aff[fv] = (float)((pqr[2] * fd[fv][0] - pqr[0] * fd[fv][2]) * sc);
                    Bff[fv] = (float)((pqr[2] * fd[fv][1] - pqr[1] * fd[fv][2]) * sc);
                }
                for (int pv = 0; pv < 3; ++pv)
                    for (int fv = 0; fv < 10; ++fv)
                        W[10 * pv + fv] += (double)Apf[pv] * (double)Aff[fv] + (double)Bpf[pv] * (double)Bff[fv];
                continue;
            }
            for (int pv = 0; pv < 3; ++pv)
                for (int fv = 0; fv < 10; ++fv)
                    W[10 * pv + fv] +=
                        second_deriv(pqr, pd[pv][0], pd[pv][1], pd[pv][2], fd[fv][0], fd[fv][1], fd[fv][2]);
        }
    }
    csc_free(&c);
    if (orc_w_f32 == 1)
        for (int64_t e = 0; e < 30 * O; ++e) Wpf[e] = (double)(float)Wpf[e];
}

/* ---------------------------------------------------------------- gauge index map */

/* BA:539-563 InitializeNormalizedVarIndices: removed frame-local variables 4..9 of frame 0 and 14+comp.
 * red[fi] = reduced index of full frame variable fi (0..10M-1), or -1 when removed. */
static int64_t* gauge_map(int32_t M, int32_t comp)
{
    int64_t nf = 10 * (int64_t)M;
    int64_t* red = (int64_t*)malloc(sizeof(int64_t) * (size_t)nf);
    int64_t r = 0;
    for (int64_t fi = 0; fi < nf; ++fi) {
        int removed = (fi >= 4 && fi <= 9) || (fi == 14 + comp);
        red[fi] = removed ? -1 : r++;
    }
    return red;
}

/* back substitution BA:1919-1960 and gap fill BA:1600-1679 (shared by the QR and the skyline-Cholesky variants) */
static int two_phase_backsub(int64_t N, int32_t M, const int64_t* row_ptr, const int32_t* obs_frame, const double* gradE,
                             const double* Vpp, const double* Wpf, double c, const int64_t* red, const double* dc,
                             double* corrections)
{
    memset(corrections, 0, sizeof(double) * (size_t)(3 * N + 10 * (int64_t)M));
    int all_finite = 1;
#pragma omp parallel for schedule(static) num_threads(orc_threads) reduction(&& : all_finite) if (orc_threads > 1)
    for (int64_t i = 0; i < N; ++i) {
        double E[9], Einv[9], det;
        memcpy(E, Vpp + 9 * i, sizeof E);
        E[0] *= 1 + c; E[4] *= 1 + c; E[8] *= 1 + c;
        double* dx = corrections + 3 * i;
        if (!orc_inverse3x3_with_check(E, Einv, &det)) { dx[0] = dx[1] = dx[2] = 0; continue; }
        const double* g = gradE + 3 * i;
        double acc[3] = { 0, 0, 0 };
        for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o)
            for (int fv = 0; fv < 10; ++fv) {
                int64_t r = red[10 * (int64_t)obs_frame[o] + fv];
                if (r < 0) continue;
                for (int pv = 0; pv < 3; ++pv) acc[pv] += Wpf[30 * o + 10 * pv + fv] * dc[r];
            }
        double b[3] = { acc[0] + g[0], acc[1] + g[1], acc[2] + g[2] };
        for (int pv = 0; pv < 3; ++pv) {
            dx[pv] = -Einv[3 * pv] * b[0] - Einv[3 * pv + 1] * b[1] - Einv[3 * pv + 2] * b[2];
            if (!isfinite(dx[pv])) all_finite = 0; /* :1953-1954 (the reference stops at the first; the result is discarded either way) */
        }
    }
    for (int64_t fi = 0; fi < 10 * (int64_t)M; ++fi) corrections[3 * N + fi] = red[fi] >= 0 ? dc[red[fi]] : 0.0;
    return all_finite;
}

int orc_two_phase_skyline(int64_t N, int32_t M, const int64_t* row_ptr, const int32_t* obs_frame, const double* gradE,
                          const double* Vpp, const double* Uff, const double* Wpf, double c, int32_t comp,
                          double* corrections, const int64_t* sel_rows, int64_t n_sel, double* S_rows_out,
                          double* rhs_out, double* sec_schur, double* sec_solve, double* sec_backsub);

/* ---------------------------------------------------------------- two-phase solve */

/* BA:1771-1995 EstimateCorrectionsDecomposedInTwoPhases (+ :1600-1679 FillCorrectionsGapsFromNormalized) */
int orc_two_phase(int64_t N, int32_t M, const int64_t* row_ptr, const int32_t* obs_frame, const double* gradE,
                  const double* Vpp, const double* Uff, const double* Wpf, double c, int32_t comp,
                  int32_t dense_literal, double* corrections, double* S_out, double* rhs_out, double* sec_schur,
                  double* sec_solve, double* sec_backsub)
{
    if (orc_solver == 1 && !dense_literal && !S_out && !rhs_out)
        return orc_two_phase_skyline(N, M, row_ptr, obs_frame, gradE, Vpp, Uff, Wpf, c, comp, corrections, NULL, 0, NULL,
                                     NULL, sec_schur, sec_solve, sec_backsub);
    int64_t n = 10 * (int64_t)M - 7;
    int64_t* red = gauge_map(M, comp);
    double t0 = now_sec();
    /* left side, column-major n x n (symmetric, so layout is immaterial for the values) */
    double* S = (double*)calloc((size_t)(n * n), sizeof(double));
    double* rhs = (double*)calloc((size_t)n, sizeof(double));
    /* fill_matG BA:1780-1823 */
    for (int32_t j = 0; j < M; ++j) {
        const double* U = Uff + 100 * (int64_t)j;
        for (int v1 = 0; v1 < 10; ++v1) {
            int64_t r1 = red[10 * (int64_t)j + v1];
            if (r1 < 0) continue;
            for (int v2 = 0; v2 < 10; ++v2) {
                int64_t r2 = red[10 * (int64_t)j + v2];
                if (r2 < 0) continue;
                double val = U[10 * v1 + v2];
                if (v1 == v2) val *= 1 + c; /* :1818-1819 */
                S[r2 * n + r1] = val;
            }
        }
    }
    double* Fd = NULL;  /* dense_literal: 3 x n row-block, column-major like Eigen (3 per column) */
    double* tmpd = NULL;
    if (dense_literal) {
        Fd = (double*)malloc(sizeof(double) * (size_t)(3 * n));
        tmpd = (double*)malloc(sizeof(double) * (size_t)(3 * n));
    }
    /* per point BA:1862-1898.  threads > 1 (sparse path only, baseline variant iii): the same terms, walked frame-major
     * so that no two threads touch the same row of S: a thread takes whole frames j, and for every observation (i, j) of
     * frame j -- in ascending landmark order, as the sequential loop meets them -- applies landmark i's updates of the ten
     * rows of frame j.  Every entry of S and rhs receives its terms in the sequential order: bit-identical results. */
    if (orc_threads > 1 && !dense_literal) {
        double* Einv_all = (double*)malloc(sizeof(double) * (size_t)(9 * (N > 0 ? N : 1)));
        char* ok_all = (char*)malloc((size_t)(N > 0 ? N : 1));
#pragma omp parallel for schedule(static) num_threads(orc_threads)
        for (int64_t i = 0; i < N; ++i) {
            double E[9], det;
            memcpy(E, Vpp + 9 * i, sizeof E);
            E[0] *= 1 + c; E[4] *= 1 + c; E[8] *= 1 + c;
            ok_all[i] = (char)orc_inverse3x3_with_check(E, Einv_all + 9 * i, &det);
        }
        csc_t cs;
        csc_build(N, M, row_ptr, obs_frame, &cs);
#pragma omp parallel for schedule(dynamic, 1) num_threads(orc_threads)
        for (int32_t j = 0; j < M; ++j) {
            for (int64_t k = cs.col_ptr[j]; k < cs.col_ptr[j + 1]; ++k) {
                const int64_t oa = cs.obs[k], i = cs.pnt[k];
                if (!ok_all[i]) continue;
                const double* Einv = Einv_all + 9 * i;
                const double* g = gradE + 3 * i;
                const double* Wa = Wpf + 30 * oa;
                double tmp[10][3];
                for (int fa = 0; fa < 10; ++fa)
                    for (int kk = 0; kk < 3; ++kk)
                        tmp[fa][kk] = Wa[fa] * Einv[0 * 3 + kk] + Wa[10 + fa] * Einv[1 * 3 + kk] + Wa[20 + fa] * Einv[2 * 3 + kk];
                for (int fa = 0; fa < 10; ++fa) {
                    int64_t ra = red[10 * (int64_t)j + fa];
                    if (ra < 0) continue;
                    for (int64_t ob = row_ptr[i]; ob < row_ptr[i + 1]; ++ob) {
                        const double* Wb = Wpf + 30 * ob;
                        for (int fb = 0; fb < 10; ++fb) {
                            int64_t rb = red[10 * (int64_t)obs_frame[ob] + fb];
                            if (rb < 0) continue;
                            S[rb * n + ra] -= tmp[fa][0] * Wb[fb] + tmp[fa][1] * Wb[10 + fb] + tmp[fa][2] * Wb[20 + fb];
                        }
                    }
                    rhs[ra] += tmp[fa][0] * g[0] + tmp[fa][1] * g[1] + tmp[fa][2] * g[2];
                }
            }
        }
        csc_free(&cs);
        free(Einv_all);
        free(ok_all);
    } else
    for (int64_t i = 0; i < N; ++i) {
        double E[9], Einv[9], det;
        memcpy(E, Vpp + 9 * i, sizeof E);
        E[0] *= 1 + c; E[4] *= 1 + c; E[8] *= 1 + c; /* :1825-1834 */
        if (!orc_inverse3x3_with_check(E, Einv, &det)) continue; /* :1877-1881 */
        const double* g = gradE + 3 * i;
        int64_t o0 = row_ptr[i], o1 = row_ptr[i + 1];
        if (dense_literal) {
            memset(Fd, 0, sizeof(double) * (size_t)(3 * n));
            for (int64_t o = o0; o < o1; ++o)
                for (int fv = 0; fv < 10; ++fv) {
                    int64_t r = red[10 * (int64_t)obs_frame[o] + fv];
                    if (r < 0) continue;
                    for (int pv = 0; pv < 3; ++pv) Fd[3 * r + pv] = Wpf[30 * o + 10 * pv + fv];
                }
            /* tmp = F^T * Einv  (n x 3) */
            for (int64_t a = 0; a < n; ++a)
                for (int k = 0; k < 3; ++k)
                    tmpd[3 * a + k] = Fd[3 * a + 0] * Einv[0 * 3 + k] + Fd[3 * a + 1] * Einv[1 * 3 + k] +
                                      Fd[3 * a + 2] * Einv[2 * 3 + k];
            /* S -= tmp * F ; rhs += tmp * g   (:1891-1897) */
            for (int64_t b = 0; b < n; ++b) {
                double f0b = Fd[3 * b], f1b = Fd[3 * b + 1], f2b = Fd[3 * b + 2];
                double* Sb = S + b * n;
                for (int64_t a = 0; a < n; ++a)
                    Sb[a] -= tmpd[3 * a] * f0b + tmpd[3 * a + 1] * f1b + tmpd[3 * a + 2] * f2b;
            }
            for (int64_t a = 0; a < n; ++a) rhs[a] += tmpd[3 * a] * g[0] + tmpd[3 * a + 1] * g[1] + tmpd[3 * a + 2] * g[2];
        } else {
            /* identical arithmetic restricted to the structurally non-zero columns */
            for (int64_t oa = o0; oa < o1; ++oa) {
                const double* Wa = Wpf + 30 * oa;
                double tmp[10][3];
                for (int fa = 0; fa < 10; ++fa)
                    for (int k = 0; k < 3; ++k)
                        tmp[fa][k] = Wa[fa] * Einv[0 * 3 + k] + Wa[10 + fa] * Einv[1 * 3 + k] +
                                     Wa[20 + fa] * Einv[2 * 3 + k];
                for (int fa = 0; fa < 10; ++fa) {
                    int64_t ra = red[10 * (int64_t)obs_frame[oa] + fa];
                    if (ra < 0) continue;
                    for (int64_t ob = o0; ob < o1; ++ob) {
                        const double* Wb = Wpf + 30 * ob;
                        for (int fb = 0; fb < 10; ++fb) {
                            int64_t rb = red[10 * (int64_t)obs_frame[ob] + fb];
                            if (rb < 0) continue;
                            S[rb * n + ra] -= tmp[fa][0] * Wb[fb] + tmp[fa][1] * Wb[10 + fb] + tmp[fa][2] * Wb[20 + fb];
                        }
                    }
                    rhs[ra] += tmp[fa][0] * g[0] + tmp[fa][1] * g[1] + tmp[fa][2] * g[2];
                }
            }
        }
    }
    /* rhs -= normalized frame derivatives BA:1902-1908 */
    for (int64_t fi = 0; fi < 10 * (int64_t)M; ++fi)
        if (red[fi] >= 0) rhs[red[fi]] -= gradE[3 * N + fi];
    if (S_out)
        for (int64_t r = 0; r < n; ++r)
            for (int64_t cc = 0; cc < n; ++cc) S_out[r * n + cc] = S[cc * n + r];
    if (rhs_out) memcpy(rhs_out, rhs, sizeof(double) * (size_t)n);
    double t1 = now_sec();
    /* BA:1911 householderQr().solve */
    double* dc = (double*)malloc(sizeof(double) * (size_t)n);
    int ok = 1;
    if (orc_skip_solve) memset(dc, 0, sizeof(double) * (size_t)n);
    else ok = orc_householder_qr_solve(n, S, rhs, dc);
    double t2 = now_sec();
    if (ok) ok = two_phase_backsub(N, M, row_ptr, obs_frame, gradE, Vpp, Wpf, c, red, dc, corrections);
    double t3 = now_sec();
    if (sec_schur) *sec_schur += t1 - t0;
    if (sec_solve) *sec_solve += t2 - t1;
    if (sec_backsub) *sec_backsub += t3 - t2;
    free(dc);
    free(Fd);
    free(tmpd);
    free(S);
    free(rhs);
    free(red);
    return ok;
}


/* ---------------------------------------------------------------- two-phase solve, baseline variant (ii)
 * the reference's arithmetic (BA:1771-1995) on block-sparse storage with the solver a CPU port
 * would use.  The reduced camera system is kept as a row skyline: row r (reduced index, frame j) starts at the first
 * reduced variable of the smallest frame that shares a landmark with j -- everything left of it is structurally zero and
 * Cholesky fill stays inside such an envelope.  The Schur sum is the loop of orc_two_phase restricted to the lower
 * triangle, term by term in the same order, so every stored entry carries the same bits as the dense S there
 * (tests/test_oracle_skyline.py).  The solve is a skyline Cholesky (the damped system is positive definite)
 * instead of BA:1911's Householder QR; a non-positive pivot or a non-finite solution returns 0 like the QR's non-finite
 * result does (:1912-1913).  sel_rows / S_rows_out: optional dump of selected rows of the system BEFORE the factorisation
 * (n_sel rows of n doubles, columns <= row filled, the rest zero) -- lets a test compare a 40 000-variable system on a
 * sample of rows without a 12.8 GB dense copy. */
int orc_two_phase_skyline(int64_t N, int32_t M, const int64_t* row_ptr, const int32_t* obs_frame, const double* gradE,
                          const double* Vpp, const double* Uff, const double* Wpf, double c, int32_t comp,
                          double* corrections, const int64_t* sel_rows, int64_t n_sel, double* S_rows_out,
                          double* rhs_out, double* sec_schur, double* sec_solve, double* sec_backsub)
{
    const int64_t n = 10 * (int64_t)M - 7;
    int64_t* red = gauge_map(M, comp);
    double t0 = now_sec();
    /* covisibility: smallest frame sharing a landmark with frame j (frame lists are ascending) */
    int32_t* mincv = (int32_t*)malloc(sizeof(int32_t) * (size_t)M);
    for (int32_t j = 0; j < M; ++j) mincv[j] = j;
    for (int64_t i = 0; i < N; ++i) {
        if (row_ptr[i + 1] == row_ptr[i]) continue;
        const int32_t f0i = obs_frame[row_ptr[i]];
        for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o)
            if (f0i < mincv[obs_frame[o]]) mincv[obs_frame[o]] = f0i;
    }
    int64_t* first = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int64_t* off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
    off[0] = 0;
    for (int64_t fi = 0; fi < 10 * (int64_t)M; ++fi) {
        const int64_t r = red[fi];
        if (r < 0) continue;
        int64_t ff = 10 * (int64_t)mincv[fi / 10];
        while (red[ff] < 0) ++ff; /* stops at fi at the latest: fi itself is kept */
        first[r] = red[ff];
        off[r + 1] = off[r] + (r - first[r] + 1);
    }
    double* L = (double*)calloc((size_t)off[n], sizeof(double));
    double* rhs = (double*)calloc((size_t)n, sizeof(double));
#define SK(r, cc) L[off[r] + ((cc) - first[r])]
    /* fill_matG BA:1780-1823, lower triangle */
    for (int32_t j = 0; j < M; ++j) {
        const double* U = Uff + 100 * (int64_t)j;
        for (int v1 = 0; v1 < 10; ++v1) {
            const int64_t r1 = red[10 * (int64_t)j + v1];
            if (r1 < 0) continue;
            for (int v2 = 0; v2 < 10; ++v2) {
                const int64_t r2 = red[10 * (int64_t)j + v2];
                if (r2 < 0 || r2 > r1) continue;
                double val = U[10 * v1 + v2];
                if (v1 == v2) val *= 1 + c; /* :1818-1819 */
                SK(r1, r2) = val;
            }
        }
    }
    /* per point BA:1862-1898, the terms of orc_two_phase in its order; threads > 1: frame-major as there (a thread owns
     * the rows of its frames) */
    if (orc_threads > 1) {
        double* Einv_all = (double*)malloc(sizeof(double) * (size_t)(9 * (N > 0 ? N : 1)));
        char* ok_all = (char*)malloc((size_t)(N > 0 ? N : 1));
#pragma omp parallel for schedule(static) num_threads(orc_threads)
        for (int64_t i = 0; i < N; ++i) {
            double E[9], det;
            memcpy(E, Vpp + 9 * i, sizeof E);
            E[0] *= 1 + c; E[4] *= 1 + c; E[8] *= 1 + c;
            ok_all[i] = (char)orc_inverse3x3_with_check(E, Einv_all + 9 * i, &det);
        }
        csc_t cs;
        csc_build(N, M, row_ptr, obs_frame, &cs);
#pragma omp parallel for schedule(dynamic, 1) num_threads(orc_threads)
        for (int32_t j = 0; j < M; ++j) {
            for (int64_t k = cs.col_ptr[j]; k < cs.col_ptr[j + 1]; ++k) {
                const int64_t oa = cs.obs[k], i = cs.pnt[k];
                if (!ok_all[i]) continue;
                const double* Einv = Einv_all + 9 * i;
                const double* g = gradE + 3 * i;
                const double* Wa = Wpf + 30 * oa;
                double tmp[10][3];
                for (int fa = 0; fa < 10; ++fa)
                    for (int kk = 0; kk < 3; ++kk)
                        tmp[fa][kk] = Wa[fa] * Einv[0 * 3 + kk] + Wa[10 + fa] * Einv[1 * 3 + kk] + Wa[20 + fa] * Einv[2 * 3 + kk];
                for (int fa = 0; fa < 10; ++fa) {
                    const int64_t ra = red[10 * (int64_t)j + fa];
                    if (ra < 0) continue;
                    for (int64_t ob = row_ptr[i]; ob < row_ptr[i + 1]; ++ob) {
                        const double* Wb = Wpf + 30 * ob;
                        for (int fb = 0; fb < 10; ++fb) {
                            const int64_t rb = red[10 * (int64_t)obs_frame[ob] + fb];
                            if (rb < 0 || rb > ra) continue;
                            SK(ra, rb) -= tmp[fa][0] * Wb[fb] + tmp[fa][1] * Wb[10 + fb] + tmp[fa][2] * Wb[20 + fb];
                        }
                    }
                    rhs[ra] += tmp[fa][0] * g[0] + tmp[fa][1] * g[1] + tmp[fa][2] * g[2];
                }
            }
        }
        csc_free(&cs);
        free(Einv_all);
        free(ok_all);
    } else
    for (int64_t i = 0; i < N; ++i) {
        double E[9], Einv[9], det;
        memcpy(E, Vpp + 9 * i, sizeof E);
        E[0] *= 1 + c; E[4] *= 1 + c; E[8] *= 1 + c; /* :1825-1834 */
        if (!orc_inverse3x3_with_check(E, Einv, &det)) continue; /* :1877-1881 */
        const double* g = gradE + 3 * i;
        const int64_t o0 = row_ptr[i], o1 = row_ptr[i + 1];
        for (int64_t oa = o0; oa < o1; ++oa) {
            const double* Wa = Wpf + 30 * oa;
            double tmp[10][3];
            for (int fa = 0; fa < 10; ++fa)
                for (int k = 0; k < 3; ++k)
                    tmp[fa][k] = Wa[fa] * Einv[0 * 3 + k] + Wa[10 + fa] * Einv[1 * 3 + k] + Wa[20 + fa] * Einv[2 * 3 + k];
            for (int fa = 0; fa < 10; ++fa) {
                const int64_t ra = red[10 * (int64_t)obs_frame[oa] + fa];
                if (ra < 0) continue;
                for (int64_t ob = o0; ob < o1; ++ob) {
                    const double* Wb = Wpf + 30 * ob;
                    for (int fb = 0; fb < 10; ++fb) {
                        const int64_t rb = red[10 * (int64_t)obs_frame[ob] + fb];
                        if (rb < 0 || rb > ra) continue;
                        SK(ra, rb) -= tmp[fa][0] * Wb[fb] + tmp[fa][1] * Wb[10 + fb] + tmp[fa][2] * Wb[20 + fb];
                    }
                }
                rhs[ra] += tmp[fa][0] * g[0] + tmp[fa][1] * g[1] + tmp[fa][2] * g[2];
            }
        }
    }
    for (int64_t fi = 0; fi < 10 * (int64_t)M; ++fi) /* BA:1902-1908 */
        if (red[fi] >= 0) rhs[red[fi]] -= gradE[3 * N + fi];
    if (S_rows_out)
        for (int64_t s = 0; s < n_sel; ++s) {
            const int64_t r = sel_rows[s];
            memset(S_rows_out + s * n, 0, sizeof(double) * (size_t)n);
            if (r < 0 || r >= n) continue;
            memcpy(S_rows_out + s * n + first[r], &SK(r, first[r]), sizeof(double) * (size_t)(r - first[r] + 1));
        }
    if (rhs_out) memcpy(rhs_out, rhs, sizeof(double) * (size_t)n);
    double t1 = now_sec();
    /* skyline Cholesky, row by row: L_ij = (a_ij - sum_k L_ik L_jk) / L_jj over the common part of rows i and j */
    double* dc = (double*)malloc(sizeof(double) * (size_t)n);
    int ok = 1;
    if (orc_skip_solve) memset(dc, 0, sizeof(double) * (size_t)n);
    else {
        for (int64_t i = 0; i < n && ok; ++i) {
            double* Li = L + off[i] - first[i];
            for (int64_t j = first[i]; j <= i; ++j) {
                const double* Lj = L + off[j] - first[j];
                const int64_t k0 = first[i] > first[j] ? first[i] : first[j];
                double sum = Li[j];
                for (int64_t k = k0; k < j; ++k) sum -= Li[k] * Lj[k];
                if (j < i) Li[j] = sum / Lj[j];
                else if (!(sum > 0) || !isfinite(sum)) { ok = 0; break; }
                else Li[i] = sqrt(sum);
            }
        }
        if (ok) {
            for (int64_t i = 0; i < n; ++i) { /* L y = rhs */
                const double* Li = L + off[i] - first[i];
                double sum = rhs[i];
                for (int64_t k = first[i]; k < i; ++k) sum -= Li[k] * dc[k];
                dc[i] = sum / Li[i];
            }
            for (int64_t i = n - 1; i >= 0; --i) { /* L^T x = y */
                const double* Li = L + off[i] - first[i];
                const double xi = dc[i] / Li[i];
                dc[i] = xi;
                for (int64_t k = first[i]; k < i; ++k) dc[k] -= Li[k] * xi;
            }
            for (int64_t i = 0; i < n; ++i)
                if (!isfinite(dc[i])) { ok = 0; break; } /* :1912-1913 */
        }
    }
#undef SK
    double t2 = now_sec();
    if (ok) ok = two_phase_backsub(N, M, row_ptr, obs_frame, gradE, Vpp, Wpf, c, red, dc, corrections);
    double t3 = now_sec();
    if (sec_schur) *sec_schur += t1 - t0;
    if (sec_solve) *sec_solve += t2 - t1;
    if (sec_backsub) *sec_backsub += t3 - t2;
    free(dc); free(L); free(rhs); free(first); free(off); free(mincv); free(red);
    return ok;
}

/* BA:1700-1769 EstimateCorrectionsNaive + BA:1551-1598 FillHessian (self-check, tiny scenes) */
int orc_naive_solve(int64_t N, int32_t M, const int64_t* row_ptr, const int32_t* obs_frame, const double* gradE,
                    const double* Vpp, const double* Uff, const double* Wpf, double c, int32_t comp,
                    double* corrections)
{
    int64_t nfull = 3 * N + 10 * (int64_t)M;
    int64_t n = nfull - 7;
    int64_t* redf = gauge_map(M, comp);
    int64_t* red = (int64_t*)malloc(sizeof(int64_t) * (size_t)nfull);
    for (int64_t i = 0; i < 3 * N; ++i) red[i] = i;
    for (int64_t fi = 0; fi < 10 * (int64_t)M; ++fi) red[3 * N + fi] = redf[fi] < 0 ? -1 : 3 * N + redf[fi];
    double* H = (double*)calloc((size_t)(n * n), sizeof(double));
    double* b = (double*)calloc((size_t)n, sizeof(double));
    for (int64_t i = 0; i < N; ++i) {
        for (int v1 = 0; v1 < 3; ++v1)
            for (int v2 = 0; v2 < 3; ++v2) {
                double val = Vpp[9 * i + 3 * v1 + v2];
                if (v1 == v2) val *= 1 + c;
                H[(3 * i + v2) * n + 3 * i + v1] = val;
            }
        for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o)
            for (int pv = 0; pv < 3; ++pv)
                for (int fv = 0; fv < 10; ++fv) {
                    int64_t r = red[3 * N + 10 * (int64_t)obs_frame[o] + fv];
                    if (r < 0) continue;
                    double val = Wpf[30 * o + 10 * pv + fv];
                    H[r * n + 3 * i + pv] = val;
                    H[(3 * i + pv) * n + r] = val;
                }
    }
    for (int32_t j = 0; j < M; ++j)
        for (int v1 = 0; v1 < 10; ++v1)
            for (int v2 = 0; v2 < 10; ++v2) {
                int64_t r1 = red[3 * N + 10 * (int64_t)j + v1], r2 = red[3 * N + 10 * (int64_t)j + v2];
                if (r1 < 0 || r2 < 0) continue;
                double val = Uff[100 * (int64_t)j + 10 * v1 + v2];
                if (v1 == v2) val *= 1 + c;
                H[r2 * n + r1] = val;
            }
    for (int64_t i = 0; i < nfull; ++i)
        if (red[i] >= 0) b[red[i]] = -gradE[i];
    double* x = (double*)malloc(sizeof(double) * (size_t)n);
    int ok = orc_householder_qr_solve(n, H, b, x);
    for (int64_t i = 0; i < nfull; ++i) corrections[i] = red[i] >= 0 ? x[red[i]] : 0.0;
    free(x);
    free(H);
    free(b);
    free(red);
    free(redf);
    return ok;
}

/* ---------------------------------------------------------------- apply */

/* BA:1997-2063 ApplyCorrections (+ :59-92 IncrementRotMat).  K is edited on a local copy in the
 * reference (:2027-2034) and dropped: intrinsics never change. */
void orc_apply_corrections(int64_t N, double* points, int32_t M, double* cam_R, double* cam_T,
                           const double* corr)
{
    for (int64_t i = 0; i < 3 * N; ++i) points[i] += corr[i];
    for (int32_t j = 0; j < M; ++j) {
        const double* d = corr + 3 * N + 10 * (int64_t)j;
        double Rd[9], Td[3];
        orc_se3_inv(cam_R + 9 * j, cam_T + 3 * j, Rd, Td);
        Td[0] += d[4]; Td[1] += d[5]; Td[2] += d[6];
        double w[3] = { d[7], d[8], d[9] };
        double rot[9], Rnew[9];
        if (orc_rot_from_axis_angle(w, rot)) mat3_mul(rot, Rd, Rnew);
        else memcpy(Rnew, Rd, sizeof Rnew);
        orc_se3_inv(Rnew, Td, cam_R + 9 * j, cam_T + 3 * j);
    }
}

/* ---------------------------------------------------------------- finite-difference derivative checkers
 * BA:895-1138 (GetFiniteDiffFirstPartialDeriv* / SecondPartialDeriv*), the patches BA:336-394 and AddDeltaToFrameInplace
 * BA:94-120: the reference's debug-only validation of the closed-form derivatives (rough_rtol 0.2, log only).  Central
 * differences of the reprojection error with ONE landmark and / or ONE frame replaced by a perturbed copy.  Frame
 * variables are perturbed as the reference perturbs them: fx fy u0 v0 on K, T and W on the DIRECT pose (T += dT,
 * R <- Rodrigues(dW) R, BA:59-92), then inverted back.  Test utility of the oracle only. */
static double fd_error(double f0, int64_t N, const double* points, int32_t M, const double* cam_R, const double* cam_T,
                       const double* K, int32_t shared_k, const int64_t* row_ptr, const int32_t* obs_frame,
                       const double* obs_uv, int64_t pi, const double* dX /* NULL = no landmark patch */, int32_t fj,
                       const double* dF /* [10], NULL = no frame patch */)
{
    double* P = (double*)malloc(sizeof(double) * (size_t)(3 * (N > 0 ? N : 1)));
    double* R = (double*)malloc(sizeof(double) * (size_t)(9 * M));
    double* T = (double*)malloc(sizeof(double) * (size_t)(3 * M));
    double* Kf = (double*)malloc(sizeof(double) * (size_t)(9 * M));
    memcpy(P, points, sizeof(double) * (size_t)(3 * N));
    memcpy(R, cam_R, sizeof(double) * (size_t)(9 * M));
    memcpy(T, cam_T, sizeof(double) * (size_t)(3 * M));
    for (int32_t j = 0; j < M; ++j) memcpy(Kf + 9 * j, shared_k ? K : K + 9 * j, sizeof(double) * 9);
    if (dX)
        for (int v = 0; v < 3; ++v) P[3 * pi + v] += dX[v]; /* SalientPointPatch BA:336-360 */
    if (dF) { /* FramePatch BA:362-394 via AddDeltaToFrameInplace BA:94-120 */
        double* Kj = Kf + 9 * fj;
        Kj[0] += dF[0]; Kj[4] += dF[1]; Kj[2] += dF[2]; Kj[5] += dF[3];
        double Rd[9], Td[3], rot[9], Rn[9];
        orc_se3_inv(R + 9 * fj, T + 3 * fj, Rd, Td);
        Td[0] += dF[4]; Td[1] += dF[5]; Td[2] += dF[6];
        double w[3] = { dF[7], dF[8], dF[9] };
        if (orc_rot_from_axis_angle(w, rot)) mat3_mul(rot, Rd, Rn);
        else memcpy(Rn, Rd, sizeof Rn);
        orc_se3_inv(Rn, Td, R + 9 * fj, T + 3 * fj);
    }
    double e = orc_reproj_error(f0, N, P, M, R, T, Kf, 0, row_ptr, obs_frame, obs_uv, NULL);
    free(P); free(R); free(T); free(Kf);
    return e;
}

#define FD_ARGS f0, N, points, M, cam_R, cam_T, K, shared_k, row_ptr, obs_frame, obs_uv
/* landmark pi: first derivatives d1[3] (BA:895-912) and second derivatives d2[3][3] (BA:914-946) */
void orc_fd_point(double f0, int64_t N, const double* points, int32_t M, const double* cam_R, const double* cam_T,
                  const double* K, int32_t shared_k, const int64_t* row_ptr, const int32_t* obs_frame,
                  const double* obs_uv, int64_t pi, double eps, double* d1, double* d2)
{
    for (int v = 0; v < 3; ++v) {
        double a[3] = { 0, 0, 0 }, b[3] = { 0, 0, 0 };
        a[v] = -eps; b[v] = eps;
        d1[v] = (fd_error(FD_ARGS, pi, b, 0, NULL) - fd_error(FD_ARGS, pi, a, 0, NULL)) / (2 * eps);
    }
    for (int v1 = 0; v1 < 3; ++v1)
        for (int v2 = 0; v2 < 3; ++v2) {
            double e[4];
            const double s1[4] = { 1, 1, -1, -1 }, s2[4] = { 1, -1, 1, -1 };
            for (int q = 0; q < 4; ++q) {
                double dx[3] = { 0, 0, 0 };
                dx[v1] += s1[q] * eps;
                dx[v2] += s2[q] * eps;
                e[q] = fd_error(FD_ARGS, pi, dx, 0, NULL);
            }
            d2[3 * v1 + v2] = (e[0] - e[1] - e[2] + e[3]) / (4 * eps * eps);
        }
}
/* frame fj: first derivatives d1[10] (BA:948-1030) and second derivatives d2[10][10] (BA:1031-1086) */
void orc_fd_frame(double f0, int64_t N, const double* points, int32_t M, const double* cam_R, const double* cam_T,
                  const double* K, int32_t shared_k, const int64_t* row_ptr, const int32_t* obs_frame,
                  const double* obs_uv, int32_t fj, double eps, double* d1, double* d2)
{
    for (int v = 0; v < 10; ++v) {
        double a[10] = { 0 }, b[10] = { 0 };
        a[v] = -eps; b[v] = eps;
        d1[v] = (fd_error(FD_ARGS, 0, NULL, fj, b) - fd_error(FD_ARGS, 0, NULL, fj, a)) / (2 * eps);
    }
    for (int v1 = 0; v1 < 10; ++v1)
        for (int v2 = 0; v2 < 10; ++v2) {
            double e[4];
            const double s1[4] = { 1, 1, -1, -1 }, s2[4] = { 1, -1, 1, -1 };
            for (int q = 0; q < 4; ++q) {
                double df[10] = { 0 };
                df[v1] += s1[q] * eps;
                df[v2] += s2[q] * eps;
                e[q] = fd_error(FD_ARGS, 0, NULL, fj, df);
            }
            d2[10 * v1 + v2] = (e[0] - e[1] - e[2] + e[3]) / (4 * eps * eps);
        }
}
/* landmark pi x frame fj: second derivatives d2[3][10] (BA:1088-1138) */
void orc_fd_point_frame(double f0, int64_t N, const double* points, int32_t M, const double* cam_R,
                        const double* cam_T, const double* K, int32_t shared_k, const int64_t* row_ptr,
                        const int32_t* obs_frame, const double* obs_uv, int64_t pi, int32_t fj, double eps, double* d2)
{
    for (int pv = 0; pv < 3; ++pv)
        for (int fv = 0; fv < 10; ++fv) {
            double e[4];
            const double s1[4] = { 1, 1, -1, -1 }, s2[4] = { 1, -1, 1, -1 };
            for (int q = 0; q < 4; ++q) {
                double dx[3] = { 0, 0, 0 }, df[10] = { 0 };
                dx[pv] = s1[q] * eps;
                df[fv] = s2[q] * eps;
                e[q] = fd_error(FD_ARGS, pi, dx, fj, df);
            }
            d2[10 * pv + fv] = (e[0] - e[1] - e[2] + e[3]) / (4 * eps * eps);
        }
}
#undef FD_ARGS

/* ---------------------------------------------------------------- LM driver */

const char* orc_status_string(int status)
{
    switch (status) {
    case ORC_STATUS_ABS_ERR_THRESHOLD: return "abs err threshold";
    case ORC_STATUS_SMALL_ERR_CHANGE: return "small relative err change";
    case ORC_STATUS_HESSIAN_OVERFLOW: return "hessian overflow";
    case ORC_STATUS_ERR_CONVERGED: return "err converged to limit value";
    case ORC_STATUS_MAX_ITERATIONS: return "max iterations";
    default: return "";
    }
}

/* BA:617-718 ComputeInplace + BA:720-893 ComputeOnNormalizedWorld */
int orc_compute_inplace(double f0, int64_t N, double* points, int32_t M, double* cam_R, double* cam_T,
                        const double* K, int32_t shared_k, const int64_t* row_ptr, const int32_t* obs_frame,
                        const double* obs_uv, const double* allowed_err_change, const double* max_hessian_factor,
                        int64_t max_iterations, int32_t dense_literal, orc_report* rep)
{
    orc_report local;
    if (!rep) rep = &local;
    memset(rep, 0, sizeof *rep);
    const int32_t comp = 1;   /* bundle-adj-kanatani.h:131-132 unity_t1_comp_ind_ = 1, value 1.0 */
    const double t1y = 1.0;
    orc_normalizer nrm;
    if (!orc_normalize_scene(N, points, M, cam_R, cam_T, t1y, comp, &nrm)) { /* :680-682 */
        rep->status = ORC_STATUS_NONE;
        rep->optimized = 0;
        return 1;
    }
    rep->world_scale = nrm.world_scale;
    int64_t O = row_ptr[N];
    int64_t nvars = 3 * N + 10 * (int64_t)M;
    double* gradE = (double*)malloc(sizeof(double) * (size_t)nvars);
    double* Vpp = (double*)malloc(sizeof(double) * (size_t)(9 * N + 1));
    double* Uff = (double*)malloc(sizeof(double) * (size_t)(100 * (int64_t)M));
    double* Wpf = (double*)malloc(sizeof(double) * (size_t)(30 * O + 1));
    double* corr = (double*)malloc(sizeof(double) * (size_t)nvars);
    double* pts_bak = (double*)malloc(sizeof(double) * (size_t)(3 * N + 1));
    double* R_bak = (double*)malloc(sizeof(double) * (size_t)(9 * (int64_t)M));
    double* T_bak = (double*)malloc(sizeof(double) * (size_t)(3 * (int64_t)M));

    double hessian_factor = (double)0.0001f; /* :723 float literal */
    int64_t seen = 0;
    double tt = now_sec();
    double err_initial = orc_reproj_error(f0, N, points, M, cam_R, cam_T, K, shared_k, row_ptr, obs_frame, obs_uv, &seen);
    rep->sec_error += now_sec() - tt;
    rep->seen = seen;
    rep->err_initial = err_initial;
    rep->err_final = err_initial;
    int result_true = 0;
    int done = 0;
    if (allowed_err_change && err_initial < *allowed_err_change) { /* :749-753 */
        rep->status = ORC_STATUS_ABS_ERR_THRESHOLD;
        result_true = 1;
        done = 1;
    }
    double err_value = err_initial;
    while (!done) {
        if (max_iterations > 0 && rep->iterations >= max_iterations) {
            rep->status = ORC_STATUS_MAX_ITERATIONS; /* harness addition: neither reference outcome */
            result_true = 0;
            break;
        }
        tt = now_sec();
        orc_derivatives(f0, N, points, M, cam_R, cam_T, K, shared_k, row_ptr, obs_frame, obs_uv, gradE, Vpp, Uff, Wpf);
        rep->sec_derivatives += now_sec() - tt;
        /* try_decrease_targ_fun :764-852 */
        memcpy(pts_bak, points, sizeof(double) * (size_t)(3 * N));
        memcpy(R_bak, cam_R, sizeof(double) * (size_t)(9 * (int64_t)M));
        memcpy(T_bak, cam_T, sizeof(double) * (size_t)(3 * (int64_t)M));
        int have_prev = 0;
        double err_new_prev = 0, err_new = NAN;
        int decrease = 0; /* 1 success, 2 overflow, 3 converged */
        while (!decrease) {
            rep->attempts += 1;
            int suc = orc_two_phase(N, M, row_ptr, obs_frame, gradE, Vpp, Uff, Wpf, hessian_factor, comp, dense_literal,
                                    corr, NULL, NULL, &rep->sec_schur, &rep->sec_solve, &rep->sec_backsub);
            if (!suc) { decrease = 2; break; } /* :807-808 */
            tt = now_sec();
            orc_apply_corrections(N, points, M, cam_R, cam_T, corr);
            rep->sec_apply += now_sec() - tt;
            tt = now_sec();
            err_new = orc_reproj_error(f0, N, points, M, cam_R, cam_T, K, shared_k, row_ptr, obs_frame, obs_uv, NULL);
            rep->sec_error += now_sec() - tt;
            if (err_new - err_value < 0) { decrease = 1; break; } /* :816-819 */
            memcpy(points, pts_bak, sizeof(double) * (size_t)(3 * N)); /* :823-826 */
            memcpy(cam_R, R_bak, sizeof(double) * (size_t)(9 * (int64_t)M));
            memcpy(cam_T, T_bak, sizeof(double) * (size_t)(3 * (int64_t)M));
            if (have_prev && allowed_err_change) { /* :828-838 */
                double change = err_new - err_new_prev;
                if (fabs(change) < *allowed_err_change) { decrease = 3; break; }
            }
            hessian_factor *= 10; /* :841 */
            if (max_hessian_factor && hessian_factor > *max_hessian_factor) { decrease = 2; break; } /* :843-847 */
            err_new_prev = err_new;
            have_prev = 1;
        }
        if (decrease != 1) { /* :857-873 */
            rep->status = decrease == 2 ? ORC_STATUS_HESSIAN_OVERFLOW : ORC_STATUS_ERR_CONVERGED;
            result_true = 0;
            break;
        }
        rep->iterations += 1;
        double change = err_new - err_value;
        rep->err_final = err_new;
        if (allowed_err_change && fabs(change) < *allowed_err_change) { /* :880-884 */
            rep->status = ORC_STATUS_SMALL_ERR_CHANGE;
            result_true = 1;
            break;
        }
        err_value = err_new;
        hessian_factor /= 10; /* :889 */
    }
    rep->hessian_factor = hessian_factor;
    rep->optimized = result_true;
    orc_revert_normalization(N, points, M, cam_R, cam_T, &nrm); /* :706 */
    free(gradE); free(Vpp); free(Uff); free(Wpf); free(corr); free(pts_bak); free(R_bak); free(T_bak);
    return result_true ? 0 : 1;
}
