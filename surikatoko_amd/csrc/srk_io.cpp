// srk_io.cpp -- host-side callers of the BA path: the dinosaur loader and its pre-processing.
//
// Restates (cpp_impl/ of whigg/surikatoko):
//   ReadMatrixFromFile                   suriko-engine/src/mat-serialization.cpp:12-87
//   DecomposeProjMat                     suriko-engine/src/obs-geom.cpp:606-677   (Kanatani-Sugaya 2010, appendix A)
//   Triangulate3DPointByLeastSquares     suriko-engine/src/obs-geom.cpp:679-727   (Eigen colPivHouseholderQr)
//   PopulateCornersPerFrame + DinoDemo   demos/demo-bundle-adj-dinosaur.cpp:24-54, 70-230
// No Eigen: 3x3 algebra is written out; the least-squares solve is a column-pivoted Householder QR of the 2n x 3
// system (Eigen's ColPivHouseholderQR algorithm: pivot on the largest remaining column norm).
#include "../../include/srk_ba.h"
#include "srk_geom.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace {

double det3(const double* A)
{
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}
bool inv3(const double* A, double* Ai)
{
    double d = det3(A);
    if (d == 0.0 || !std::isfinite(d)) return false;
    double id = 1 / d;
    double t[9];
    t[0] = (A[4] * A[8] - A[5] * A[7]) * id;
    t[1] = (A[2] * A[7] - A[1] * A[8]) * id;
    t[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    t[3] = (A[5] * A[6] - A[3] * A[8]) * id;
    t[4] = (A[0] * A[8] - A[2] * A[6]) * id;
    t[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    t[6] = (A[3] * A[7] - A[4] * A[6]) * id;
    t[7] = (A[1] * A[6] - A[0] * A[7]) * id;
    t[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    std::memcpy(Ai, t, sizeof t);
    return true;
}
// lower Cholesky factor of a 3x3 SPD matrix; false on a non-positive pivot (Eigen LLT NumericalIssue)
bool chol3(const double* A, double* L)
{
    std::memset(L, 0, 72);
    for (int j = 0; j < 3; ++j) {
        double s = A[3 * j + j];
        for (int k = 0; k < j; ++k) s -= L[3 * j + k] * L[3 * j + k];
        if (!(s > 0)) return false;
        L[3 * j + j] = std::sqrt(s);
        for (int i = j + 1; i < 3; ++i) {
            double t = A[3 * i + j];
            for (int k = 0; k < j; ++k) t -= L[3 * i + k] * L[3 * j + k];
            L[3 * i + j] = t / L[3 * j + j];
        }
    }
    return true;
}

void set_err(char* err, int errlen, const std::string& msg)
{
    if (err && errlen > 0) {
        std::snprintf(err, (size_t)errlen, "%s", msg.c_str());
    }
}

} // namespace

extern "C" {

// mat-serialization.cpp:12-87.  data == NULL: only count.  returns 1 on success, 0 on failure (err filled).
int srk_read_matrix_file(const char* path, char delimiter, double* data, int64_t capacity, int64_t* rows, int64_t* cols,
                         char* err, int errlen)
{
    std::ifstream fs(path);
    if (!fs) {
        set_err(err, errlen, std::string("Can't open file ") + path);
        return 0;
    }
    std::string line;
    int64_t num_rows = 0, num_cols = -1, count = 0;
    while (std::getline(fs, line)) {
        int64_t cur = 0;
        size_t pos = 0;
        while (pos <= line.size()) {
            // strtok semantics: runs of the delimiter separate tokens, empty tokens are skipped
            size_t start = line.find_first_not_of(delimiter, pos);
            if (start == std::string::npos) break;
            size_t end = line.find(delimiter, start);
            if (end == std::string::npos) end = line.size();
            std::string tok = line.substr(start, end - start);
            std::istringstream iss(tok);
            double num;
            iss >> num;
            if (iss.fail() || !iss.eof()) { // the whole token must be a number (:58-66)
                set_err(err, errlen, "Can't parse number (" + tok + ") on line " + std::to_string(num_rows));
                return 0;
            }
            if (data) {
                if (count >= capacity) {
                    set_err(err, errlen, "buffer too small");
                    return 0;
                }
                data[count] = num;
            }
            ++count;
            ++cur;
            pos = end + 1;
        }
        if (num_cols == -1) num_cols = cur;
        else if (num_cols != cur) { // :72-81
            set_err(err, errlen, "Data has inconsistent number of columns, row(0).columns=" + std::to_string(num_cols) +
                                     ", row(" + std::to_string(num_rows) + ").columns=" + std::to_string(cur));
            return 0;
        }
        ++num_rows;
    }
    *rows = num_rows;
    *cols = num_cols == -1 ? 0 : num_cols;
    return 1;
}

// obs-geom.cpp:606-677.  P row-major 3x4 -> P = scale * K * R^T [I | -t]; (R, t) is the DIRECT camera pose.
int srk_decompose_proj_mat(const double* P, double* scale_factor, double* K, double* R_direct, double* T_direct)
{
    double Q[9], q[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Q[3 * r + c] = P[4 * r + c];
        q[r] = P[4 * r + 3];
    }
    int P_sign = 1;
    if (det3(Q) < 0) { // :618-624 ensure det(R) > 0
        P_sign = -1;
        for (double& v : Q) v = -v;
        for (double& v : q) v = -v;
    }
    double Qi[9], t[3];
    if (!inv3(Q, Qi)) return 0;
    srk::mat3_vec(Qi, q, t);
    for (double& v : t) v = -v; // :627-628
    double Qt[9], QQt[9], QQti[9], L[9];
    srk::mat3_tr(Q, Qt);
    srk::mat3_mul(Q, Qt, QQt);
    if (!inv3(QQt, QQti)) return 0;
    if (!chol3(QQti, L)) return 0; // :637-642
    double C[9];
    srk::mat3_tr(L, C); // upper triangular
    double CQ[9], R[9];
    srk::mat3_mul(C, Q, CQ);
    srk::mat3_tr(CQ, R); // :649
    double Ci[9];
    if (!inv3(C, Ci)) return 0;
    double c_last = Ci[8];
    if (srk::is_close(0.0, c_last)) return 0; // :660-661
    for (int i = 0; i < 9; ++i) K[i] = Ci[i] * (1 / c_last);
    *scale_factor = P_sign * c_last;
    std::memcpy(R_direct, R, sizeof R);
    std::memcpy(T_direct, t, sizeof t);
    return 1;
}

// obs-geom.cpp:679-727.  uv [n][2] pixels, P [n][12] row-major 3x4, -> X[3].  returns 1 ok, 0 bad args / rank loss.
int srk_triangulate_least_squares(int32_t n, const double* uv, const double* P, double f0, double* X)
{
    if (n < 2 || !uv || !P || !X) return 0; // CHECK(frames_count >= 2)
    const int m = 2 * n;
    std::vector<double> A((size_t)(3 * m)), B((size_t)m);
    for (int f = 0; f < n; ++f) {
        double x = uv[2 * f], y = uv[2 * f + 1];
        const double* p = P + 12 * (int64_t)f;
        for (int c = 0; c < 3; ++c) {
            A[(size_t)(3 * (2 * f) + c)] = x * p[8 + c] - f0 * p[c];
            A[(size_t)(3 * (2 * f + 1) + c)] = y * p[8 + c] - f0 * p[4 + c];
        }
        B[(size_t)(2 * f)] = -(x * p[11] - f0 * p[3]);
        B[(size_t)(2 * f + 1)] = -(y * p[11] - f0 * p[7]);
    }
    // column-pivoted Householder QR (3 columns)
    int perm[3] = { 0, 1, 2 };
    for (int k = 0; k < 3; ++k) {
        int best = k;
        double bestn = -1;
        for (int c = k; c < 3; ++c) {
            double s = 0;
            for (int r = k; r < m; ++r) s += A[(size_t)(3 * r + c)] * A[(size_t)(3 * r + c)];
            if (s > bestn) { bestn = s; best = c; }
        }
        if (!(bestn > 0)) return 0;
        if (best != k) {
            for (int r = 0; r < m; ++r) std::swap(A[(size_t)(3 * r + k)], A[(size_t)(3 * r + best)]);
            std::swap(perm[k], perm[best]);
        }
        double c0 = A[(size_t)(3 * k + k)], tail = 0;
        for (int r = k + 1; r < m; ++r) tail += A[(size_t)(3 * r + k)] * A[(size_t)(3 * r + k)];
        double beta = std::sqrt(c0 * c0 + tail);
        if (c0 >= 0) beta = -beta;
        double tau = (beta - c0) / beta, inv = c0 - beta;
        std::vector<double> v((size_t)m, 0.0);
        v[(size_t)k] = 1;
        for (int r = k + 1; r < m; ++r) v[(size_t)r] = A[(size_t)(3 * r + k)] / inv;
        for (int c = k; c < 3; ++c) {
            double s = 0;
            for (int r = k; r < m; ++r) s += v[(size_t)r] * A[(size_t)(3 * r + c)];
            s *= tau;
            for (int r = k; r < m; ++r) A[(size_t)(3 * r + c)] -= s * v[(size_t)r];
        }
        double s = 0;
        for (int r = k; r < m; ++r) s += v[(size_t)r] * B[(size_t)r];
        s *= tau;
        for (int r = k; r < m; ++r) B[(size_t)r] -= s * v[(size_t)r];
    }
    double y[3];
    for (int i = 2; i >= 0; --i) {
        double s = B[(size_t)i];
        for (int c = i + 1; c < 3; ++c) s -= A[(size_t)(3 * i + c)] * y[c];
        y[i] = s / A[(size_t)(3 * i + i)];
    }
    for (int i = 0; i < 3; ++i) X[perm[i]] = y[i];
    return std::isfinite(X[0]) && std::isfinite(X[1]) && std::isfinite(X[2]) ? 1 : 0;
}

// DinoDemo's scene construction (demo-bundle-adj-dinosaur.cpp:85-230) from the two oxfvisgeom files:
//   <dir>/dinoPs_as_mat108x4.txt   3 tab-separated rows of 4 per frame (row-major 3x4 projection matrices)
//   <dir>/viff.xy                  one row per point, x y per frame, -1 = not seen
// Two-call protocol: with points == NULL only the sizes are returned.  Tracks without corners are dropped (:45).
// K = diag(1/f0, 1/f0, 1) * K_decomposed with K(0,1) := 0 (:155-158); cameras are inverse poses; points are
// triangulated from the f0-scaled projection matrices (:172-196).
int srk_dino_load(const char* dir, double f0, int64_t* n_points, int32_t* n_frames, int64_t* n_obs, double* points,
                  double* cam_R, double* cam_T, double* K, int64_t* row_ptr, int32_t* obs_frame, double* obs_uv, char* err,
                  int errlen)
{
    std::string d(dir);
    std::string pfile = d + "/dinoPs_as_mat108x4.txt", vfile = d + "/viff.xy";
    int64_t pr = 0, pc = 0, vr = 0, vc = 0;
    if (!srk_read_matrix_file(pfile.c_str(), '\t', nullptr, 0, &pr, &pc, err, errlen)) return 0;
    if (!srk_read_matrix_file(vfile.c_str(), ' ', nullptr, 0, &vr, &vc, err, errlen)) return 0;
    if (pc != 4 || pr % 3 != 0) { set_err(err, errlen, "projection file must have 4 columns and 3 rows per frame"); return 0; }
    std::vector<double> Pd((size_t)(pr * pc)), Vd((size_t)(vr * vc));
    if (!srk_read_matrix_file(pfile.c_str(), '\t', Pd.data(), (int64_t)Pd.size(), &pr, &pc, err, errlen)) return 0;
    if (!srk_read_matrix_file(vfile.c_str(), ' ', Vd.data(), (int64_t)Vd.size(), &vr, &vc, err, errlen)) return 0;
    int32_t M = (int32_t)(pr / 3);
    if (M != vc / 2) { set_err(err, errlen, "Inconsistent frames_count"); return 0; } // :110-114
    // PopulateCornersPerFrame :24-54
    int64_t N = 0, O = 0;
    for (int64_t p = 0; p < vr; ++p) {
        int64_t cnt = 0;
        for (int32_t f = 0; f < M; ++f) {
            double x = Vd[(size_t)(p * vc + 2 * f)], y = Vd[(size_t)(p * vc + 2 * f + 1)];
            if (x == -1 || y == -1) continue;
            ++cnt;
        }
        if (cnt == 0) continue;
        ++N;
        O += cnt;
    }
    *n_points = N;
    *n_frames = M;
    *n_obs = O;
    if (!points) return 1;
    std::vector<double> Pf0((size_t)(12 * (int64_t)M));
    for (int32_t f = 0; f < M; ++f) {
        double scale, Kd[9], Rd[9], Td[3];
        if (!srk_decompose_proj_mat(&Pd[(size_t)(12 * (int64_t)f)], &scale, Kd, Rd, Td)) {
            set_err(err, errlen, "Can't decompose projection matrix for frame_ind=" + std::to_string(f));
            return 0;
        }
        double* Kn = K + 9 * (int64_t)f;
        for (int c = 0; c < 3; ++c) {
            Kn[c] = Kd[c] / f0;
            Kn[3 + c] = Kd[3 + c] / f0;
            Kn[6 + c] = Kd[6 + c];
        }
        Kn[1] = 0; // zero_cam_intrinsic_mat_01 (:156-157)
        double* R = cam_R + 9 * (int64_t)f;
        double* T = cam_T + 3 * (int64_t)f;
        srk::se3_inv(Rd, Td, R, T); // :169
        double KR[9], KT[3];
        srk::mat3_mul(Kn, R, KR);
        srk::mat3_vec(Kn, T, KT);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) Pf0[(size_t)(12 * (int64_t)f + 4 * r + c)] = KR[3 * r + c];
            Pf0[(size_t)(12 * (int64_t)f + 4 * r + 3)] = KT[r];
        }
    }
    int64_t pi = 0, o = 0;
    row_ptr[0] = 0;
    std::vector<double> uvs, Ps;
    for (int64_t p = 0; p < vr; ++p) {
        uvs.clear();
        Ps.clear();
        int64_t o0 = o;
        for (int32_t f = 0; f < M; ++f) {
            double x = Vd[(size_t)(p * vc + 2 * f)], y = Vd[(size_t)(p * vc + 2 * f + 1)];
            if (x == -1 || y == -1) continue;
            obs_frame[o] = f;
            obs_uv[2 * o] = x;
            obs_uv[2 * o + 1] = y;
            ++o;
            uvs.push_back(x);
            uvs.push_back(y);
            Ps.insert(Ps.end(), &Pf0[(size_t)(12 * (int64_t)f)], &Pf0[(size_t)(12 * (int64_t)f)] + 12);
        }
        if (o == o0) continue;
        int32_t nf = (int32_t)(o - o0);
        double X[3] = { 0, 0, 0 };
        if (nf >= 2) {
            if (!srk_triangulate_least_squares(nf, uvs.data(), Ps.data(), f0, X)) {
                set_err(err, errlen, "triangulation failed for point " + std::to_string(p));
                return 0;
            }
        } else {
            set_err(err, errlen, "point " + std::to_string(p) + " is seen in one frame only (Provide 2 or more projections)");
            return 0;
        }
        points[3 * pi] = X[0];
        points[3 * pi + 1] = X[1];
        points[3 * pi + 2] = X[2];
        ++pi;
        row_ptr[pi] = o;
    }
    return 1;
}

} // extern "C"
