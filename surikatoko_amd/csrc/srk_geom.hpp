// srk_geom.hpp -- tiny fixed-size host linear algebra for the product's host side (the reference uses Eigen,
// which is not available here).  Mirrors the geometry helpers the BA path calls:
//   SE3Inv / SE3Apply / SE3Compose / SE3AFromB      cpp_impl/suriko-engine/src/obs-geom.cpp:117-150
//   RotMatFromUnityDirAndAngle / RotMatFromAxisAngle obs-geom.cpp:520-561
//   LogSO3                                           obs-geom.cpp:563-593
//   IsClose                                          include/suriko/approx-alg.h:8-16
#pragma once
#include <cmath>
#include <cstring>

namespace srk {

inline bool is_close(double a, double b, double rtol = 1.0e-5, double atol = 1.0e-8)
{
    double mx = a > b ? a : b; // sic: max before abs, as in the reference
    return std::fabs(a - b) <= (atol + rtol * std::fabs(mx));
}

inline void mat3_mul(const double* A, const double* B, double* C)
{
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    std::memcpy(C, t, sizeof t);
}
inline void mat3_tr(const double* A, double* At)
{
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * j + i];
    std::memcpy(At, t, sizeof t);
}
inline void mat3_vec(const double* A, const double* x, double* y)
{
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
    y[0] = t[0]; y[1] = t[1]; y[2] = t[2];
}
inline double norm3(const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
inline void cross3(const double* a, const double* b, double* c)
{
    double t[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
    c[0] = t[0]; c[1] = t[1]; c[2] = t[2];
}
inline void se3_inv(const double* R, const double* T, double* Ri, double* Ti)
{
    double Rt[9], t[3];
    mat3_tr(R, Rt);
    mat3_vec(Rt, T, t);
    std::memcpy(Ri, Rt, sizeof Rt);
    Ti[0] = -t[0]; Ti[1] = -t[1]; Ti[2] = -t[2];
}
inline void se3_apply(const double* R, const double* T, const double* x, double* y)
{
    double t[3];
    mat3_vec(R, x, t);
    y[0] = t[0] + T[0]; y[1] = t[1] + T[1]; y[2] = t[2] + T[2];
}
inline bool rot_from_unity_dir_and_angle(const double* dir, double ang, double* R, bool check_input = true)
{
    if (check_input) {
        if (!is_close(1.0, norm3(dir))) return false;
        if (is_close(0.0, ang)) return false;
    }
    double s = std::sin(ang), c = std::cos(ang);
    double K[9] = { 0, -dir[2], dir[1], dir[2], 0, -dir[0], -dir[1], dir[0], 0 };
    double KK[9];
    mat3_mul(K, K, KK);
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + s * K[i] + (1 - c) * KK[i];
    return true;
}
inline bool rot_from_axis_angle(const double* w, double* R)
{
    double ang = norm3(w);
    if (is_close(0.0, ang)) return false;
    double d[3] = { w[0] / ang, w[1] / ang, w[2] / ang };
    return rot_from_unity_dir_and_angle(d, ang, R, false);
}
inline bool log_so3(const double* R, double* dir, double* ang)
{
    double cos_ang = 0.5 * (R[0] + R[4] + R[8] - 1);
    cos_ang = cos_ang < -1 ? -1 : (cos_ang > 1 ? 1 : cos_ang);
    double sin_ang = std::sqrt(1.0 - cos_ang * cos_ang);
    if (is_close(0.0, sin_ang, 0.0, (double)1e-3f)) return false;
    dir[0] = R[7] - R[5];
    dir[1] = R[2] - R[6];
    dir[2] = R[3] - R[1];
    double k = 0.5 / sin_ang;
    for (int i = 0; i < 3; ++i) dir[i] *= k;
    double il = 1 / norm3(dir);
    for (int i = 0; i < 3; ++i) dir[i] *= il;
    *ang = std::acos(cos_ang);
    return true;
}

} // namespace srk
