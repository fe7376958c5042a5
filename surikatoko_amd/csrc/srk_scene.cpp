// srk_scene.cpp -- synthetic circle-grid scenes (host only), the inputs of the benchmark configurations.
//
// Restates the scene construction of whigg/surikatoko
//   cpp_impl/demos/demo-bundle-adj-circle-grid.cpp:86-257   (grid of points, K, noise recipe, projections)
//   cpp_impl/suriko-engine/src/virt-world/scene-generator.cpp:9-55  (GenerateCircleCameraShots)
// with two documented extensions for the large configurations:
//   * visibility window: point i is seen by the L consecutive frames starting at
//     s_i = (i * 2654435761 mod 2^32) mod (M - L + 1)   (the demo projects every point into every frame);
//   * two-ring rig: odd frames fly at 0.8 * ascent.  The gauge fixes the y component of the cam0->cam1
//     translation to 1 (bundle-adj-kanatani.cpp:208-219); on a single ring that component shrinks with the
//     angular step (~R * step^2), the world scale explodes and every 3x3 point block falls under the
//     reference's absolute invertibility threshold (|det| <= 1e-12, :1876-1881).  Two rings keep it O(1).
#include "../../include/srk_ba.h"
#include "srk_geom.hpp"

#include <cmath>
#include <random>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

extern "C" {

void srk_circle_camera_shots(const double center[3], double radius, double ascent_z, int32_t n, const double* angles,
                             double* cam_R, double* cam_T)
{
    for (int32_t a = 0; a < n; ++a) {
        double ang = angles[a];
        double c2c[3] = { radius * std::cos(ang), radius * std::sin(ang), ascent_z };
        double shift[3] = { center[0] + c2c[0], center[1] + c2c[1], center[2] + c2c[2] };
        double R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        double T[3] = { -shift[0], -shift[1], -shift[2] }; // world -> camera, scene-generator.cpp:29
        double d[3] = { -shift[0], -shift[1], 0 };         // :32-33 direction towards the centre
        double dl = srk::norm3(d);
        for (int i = 0; i < 3; ++i) d[i] /= dl;
        double oy[3] = { 0, 1, 0 }, oz[3] = { 0, 0, 1 }, ox[3] = { 1, 0, 0 };
        double yaw = std::acos(oy[0] * d[0] + oy[1] * d[1] + oy[2] * d[2]); // :35
        double cr[3];
        srk::cross3(oy, d, cr);
        double dotz = cr[0] * oz[0] + cr[1] * oz[1] + cr[2] * oz[2];
        yaw *= (dotz >= 0 ? 1 : -1); // :38-40 Sign()
        double Rz[9], Rx[9], I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        if (!srk::rot_from_unity_dir_and_angle(oz, -yaw, Rz)) std::memcpy(Rz, I, sizeof I); // :42, RotMat
        srk::mat3_mul(Rz, R, R);
        srk::mat3_vec(Rz, T, T);
        double look_down = std::atan2(c2c[2], std::sqrt(c2c[0] * c2c[0] + c2c[1] * c2c[1])); // :45
        if (!srk::rot_from_unity_dir_and_angle(ox, look_down + M_PI / 2, Rx)) std::memcpy(Rx, I, sizeof I); // :48
        srk::mat3_mul(Rx, R, R);
        srk::mat3_vec(Rx, T, T);
        std::memcpy(cam_R + 9 * (int64_t)a, R, sizeof R);
        std::memcpy(cam_T + 3 * (int64_t)a, T, sizeof T);
    }
}

static int window_len(const srk_scene_spec* s)
{
    if (s->vis_window <= 0 || s->vis_window >= s->n_frames) return s->n_frames;
    return s->vis_window;
}

int64_t srk_scene_num_observations(const srk_scene_spec* s)
{
    if (!s || s->n_frames < 2 || s->grid_nx < 1 || s->grid_ny < 1) return -1;
    return (int64_t)s->grid_nx * s->grid_ny * window_len(s);
}

int srk_scene_generate(const srk_scene_spec* s, double* points, double* points_gt, double* cam_R, double* cam_T,
                       double* cam_R_gt, double* cam_T_gt, double* K, int64_t* row_ptr, int32_t* obs_frame,
                       double* obs_uv)
{
    if (!s || s->n_frames < 2 || s->grid_nx < 1 || s->grid_ny < 1 || !points || !cam_R || !cam_T || !K || !row_ptr ||
        !obs_frame || !obs_uv)
        return SRK_E_ARGS;
    const int32_t M = s->n_frames;
    const int64_t N = (int64_t)s->grid_nx * s->grid_ny;
    const int L = window_len(s);
    const double f0 = s->f0;
    const double hx = s->half_extent_x, hy = s->half_extent_y;

    // points: demo :97-107 (z = z_min + cos((x - xmid)/xlen * pi) * zlen with z in [0,1])
    std::vector<double> gt((size_t)(3 * N));
    {
        double xmid = 0.0, xlen = 2 * hx;
        int64_t i = 0;
        for (int ix = 0; ix < s->grid_nx; ++ix) {
            double x = s->grid_nx > 1 ? -hx + 2 * hx * ix / (s->grid_nx - 1) : 0.0;
            for (int iy = 0; iy < s->grid_ny; ++iy) {
                double y = s->grid_ny > 1 ? -hy + 2 * hy * iy / (s->grid_ny - 1) : 0.0;
                gt[(size_t)(3 * i)] = x;
                gt[(size_t)(3 * i + 1)] = y;
                gt[(size_t)(3 * i + 2)] = 0.0 + std::cos((x - xmid) / xlen * M_PI) * 1.0;
                ++i;
            }
        }
    }
    std::mt19937 gen(s->seed); // demo :109-111 (seed 1234)
    for (int64_t i = 0; i < 3 * N; ++i) points[i] = gt[(size_t)i];
    if (s->noise_x3d_hi > 0) { // demo :114-127
        std::uniform_real_distribution<double> dis(s->noise_x3d_hi / 2, s->noise_x3d_hi);
        for (int64_t i = 0; i < N; ++i) {
            double d1 = dis(gen), d2 = dis(gen), d3 = dis(gen);
            points[3 * i] += d1;
            points[3 * i + 1] += d2;
            points[3 * i + 2] += d3;
        }
    }
    if (points_gt) std::memcpy(points_gt, gt.data(), sizeof(double) * (size_t)(3 * N));

    // K = diag(1/f0, 1/f0, 1) * [[880,0,400],[0,660,300],[0,0,1]]  (demo :151-163)
    for (int32_t j = 0; j < M; ++j) {
        double* k = K + 9 * (int64_t)j;
        k[0] = (1 / f0) * 880; k[1] = 0; k[2] = (1 / f0) * 400;
        k[3] = 0; k[4] = (1 / f0) * 660; k[5] = (1 / f0) * 300;
        k[6] = 0; k[7] = 0; k[8] = 1;
    }

    // cameras: M angles over [-pi/3, 2pi/3) (demo :57-59,165-178), radius 15*cell, ascent 10*cell with cell = 0.5*hx
    std::vector<double> Rgt((size_t)(9 * (int64_t)M)), Tgt((size_t)(3 * (int64_t)M));
    {
        double center[3] = { 1, 0.5, 0 };
        double radius = 7.5 * hx, ascent = 5.0 * hx;
        for (int32_t k = 0; k < M; ++k) {
            double ang = -M_PI / 3 + k * (M_PI / M);
            double asc = (k % 2 == 0) ? ascent : 0.8 * ascent; // two-ring rig, see header
            srk_circle_camera_shots(center, radius, asc, 1, &ang, &Rgt[(size_t)(9 * (int64_t)k)],
                                    &Tgt[(size_t)(3 * (int64_t)k)]);
        }
    }
    if (cam_R_gt) std::memcpy(cam_R_gt, Rgt.data(), sizeof(double) * Rgt.size());
    if (cam_T_gt) std::memcpy(cam_T_gt, Tgt.data(), sizeof(double) * Tgt.size());

    // observations: exact projections of the noise-free scene (demo :31-41,196-207), point-major CSR
    std::normal_distribution<double> pixn(0.0, s->noise_uv_pix > 0 ? s->noise_uv_pix : 1.0);
    std::mt19937 gen_uv(s->seed ^ 0x9e3779b9u);
    int64_t o = 0;
    row_ptr[0] = 0;
    for (int64_t i = 0; i < N; ++i) {
        int32_t start = 0;
        if (L < M) start = (int32_t)(((uint32_t)((uint64_t)i * 2654435761ull)) % (uint32_t)(M - L + 1));
        for (int32_t j = start; j < start + L; ++j) {
            const double* R = &Rgt[(size_t)(9 * (int64_t)j)];
            const double* T = &Tgt[(size_t)(3 * (int64_t)j)];
            const double* k = K + 9 * (int64_t)j;
            double xc[3];
            srk::se3_apply(R, T, &gt[(size_t)(3 * i)], xc);
            double img[3] = { xc[0] / xc[2], xc[1] / xc[2], 1.0 };
            double pix[3];
            srk::mat3_vec(k, img, pix);
            double u = pix[0] / pix[2] * f0, v = pix[1] / pix[2] * f0;
            if (s->noise_uv_pix > 0) {
                u += pixn(gen_uv);
                v += pixn(gen_uv);
            }
            obs_frame[o] = j;
            obs_uv[2 * o] = u;
            obs_uv[2 * o + 1] = v;
            ++o;
        }
        row_ptr[i + 1] = o;
    }

    // camera noise: LogSO3, angle += U(0,1)*hi, axis += U(0,1)*hi per component, renormalise (demo :224-257)
    std::memcpy(cam_R, Rgt.data(), sizeof(double) * Rgt.size());
    std::memcpy(cam_T, Tgt.data(), sizeof(double) * Tgt.size());
    if (s->noise_r_hi > 0) {
        std::uniform_real_distribution<double> dis(0, 1);
        for (int32_t j = 0; j < M; ++j) {
            double* R = cam_R + 9 * (int64_t)j;
            double dir[3], ang;
            if (!srk::log_so3(R, dir, &ang)) continue;
            double da = dis(gen) * s->noise_r_hi;
            ang += da;
            double dw1 = dis(gen) * s->noise_r_hi, dw2 = dis(gen) * s->noise_r_hi, dw3 = dis(gen) * s->noise_r_hi;
            dir[0] += dw1; dir[1] += dw2; dir[2] += dw3;
            double il = 1 / srk::norm3(dir);
            for (int t = 0; t < 3; ++t) dir[t] *= il;
            double Rn[9];
            if (srk::rot_from_unity_dir_and_angle(dir, ang, Rn)) std::memcpy(R, Rn, sizeof Rn);
        }
    }
    return SRK_OK;
}

} // extern "C"
