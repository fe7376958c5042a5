// demo-bundle-adj-circle-grid -- drop-in of the reference demo (cpp_impl/demos/demo-bundle-adj-circle-grid.cpp:64-297)
// on the MI355X BA core: same flags (:46-62), same scene (grid of points on a cosine surface, cameras on a circle,
// mt19937(1234) noise), same call: BundleAdjustmentKanatani::ComputeInplace in per-frame-K mode (:285-293).
// No OpenCV window, no glog: the LOG(INFO) lines go to stderr.
#include <cmath>
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "flags.hpp"
#include "scene_dump.hpp"
#include "suriko_amd/bundle-adj-kanatani.hpp"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
using namespace suriko_amd;

int main(int argc, char** argv)
{
    Flags fl;
    fl.Parse(argc, argv);
    const double f0 = fl.Double("f0", 600);
    const double xmin = fl.Double("world_xmin", -1), xmax = fl.Double("world_xmax", 1);
    const double ymin = fl.Double("world_ymin", -1), ymax = fl.Double("world_ymax", 1);
    const double zmin = fl.Double("world_zmin", 0), zmax = fl.Double("world_zmax", 1);
    const double cx = fl.Double("world_cell_size_x", 0.5), cy = fl.Double("world_cell_size_y", 0.5);
    const double ang_start = fl.Double("ang_start", -M_PI / 2 + M_PI / 6), ang_end = fl.Double("ang_end", 2 * M_PI / 3);
    const double ang_step = fl.Double("ang_step", M_PI / 180 * 5);
    const double noise_R_hi = fl.Double("noise_R_hi", 0.005), noise_x3D_hi = fl.Double("noise_x3D_hi", 0.005);
    const double allowed_repr_err = fl.Double("allowed_repr_err", 1e-5);
    const long max_iterations = fl.Int("max_iterations", 0); // harness addition: the reference has no cap
    // harness additions: the scene handed to / returned by ComputeInplace, for the oracle comparison in tests/
    const std::string dump_before = fl.String("dump_scene_before", ""), dump_after = fl.String("dump_scene_after", "");
    std::fprintf(stderr, "noise_x3D_hi=%g\nnoise_R_hi=%g\n", noise_x3D_hi, noise_R_hi);

    const double rot_radius = 15 * cx, ascentZ = 10 * cx; // :86-87
    const double inclusive_gap = 1e-8;
    FragmentMap map, map_noise;
    std::vector<Point3> gt;
    double xmid = (xmin + xmax) / 2, xlen = xmax - xmin, zlen = zmax - zmin;
    for (double x = xmin; x < xmax + inclusive_gap; x += cx)      // :97-107
        for (double y = ymin; y < ymax + inclusive_gap; y += cy) {
            double z = zmin + std::cos((x - xmid) / xlen * M_PI) * zlen;
            gt.push_back({ x, y, z });
        }
    std::mt19937 gen(1234);                                        // :109-111
    std::vector<Point3> noisy = gt;
    if (noise_x3D_hi > 0) {                                        // :114-127
        std::uniform_real_distribution<double> dis(noise_x3D_hi / 2, noise_x3D_hi);
        for (Point3& p : noisy) { double d1 = dis(gen), d2 = dis(gen), d3 = dis(gen); p.x += d1; p.y += d2; p.z += d3; }
    }
    std::fprintf(stderr, "points_count=%zu\n", noisy.size());
    CornerTrackRepository track_rep;
    for (const Point3& p : noisy) {
        size_t id = map_noise.AddSalientPoint(p);
        track_rep.AddCornerTrackObj().SalientPointId = id;
    }
    Matrix3 K{ 880 / f0, 0, 400 / f0, 0, 660 / f0, 300 / f0, 0, 0, 1 }; // :151-163
    std::vector<double> angles;
    for (double ang = ang_start;; ang += ang_step) {               // :165-174
        if ((ang_start < ang_end && ang >= ang_end) || (ang_start > ang_end && ang <= ang_end)) break;
        angles.push_back(ang);
    }
    std::fprintf(stderr, "frames_count=%zu\n", angles.size());
    const size_t M = angles.size();
    std::vector<double> R(9 * M), T(3 * M);
    double center[3] = { 1, 0.5, 0 };
    srk_circle_camera_shots(center, rot_radius, ascentZ, (int32_t)M, angles.data(), R.data(), T.data()); // :178
    std::vector<SE3Transform> gt_cams(M);
    std::vector<Matrix3> Ks(M, K);
    for (size_t j = 0; j < M; ++j) {
        for (int e = 0; e < 9; ++e) gt_cams[j].R[(size_t)e] = R[9 * j + (size_t)e];
        gt_cams[j].T = { T[3 * j], T[3 * j + 1], T[3 * j + 2] };
        for (size_t i = 0; i < gt.size(); ++i) {                   // exact projections of the noise-free scene :196-207
            const Point3& X = gt[i];
            const Matrix3& r = gt_cams[j].R;
            double xc = r[0] * X.x + r[1] * X.y + r[2] * X.z + gt_cams[j].T.x;
            double yc = r[3] * X.x + r[4] * X.y + r[5] * X.z + gt_cams[j].T.y;
            double zc = r[6] * X.x + r[7] * X.y + r[8] * X.z + gt_cams[j].T.z;
            double u = (K[0] * xc / zc + K[1] * yc / zc + K[2]) * f0, v = (K[4] * yc / zc + K[5]) * f0;
            track_rep.CornerTracks[i].AddCorner(j, { u, v });
        }
    }
    std::vector<SE3Transform> cams = gt_cams;
    if (noise_R_hi > 0) {                                          // :224-257 LogSO3, perturb, back
        std::uniform_real_distribution<double> dis(0, 1);
        for (SE3Transform& rt : cams) {
            const Matrix3& m = rt.R;
            double cos_ang = 0.5 * (m[0] + m[4] + m[8] - 1);
            cos_ang = cos_ang < -1 ? -1 : (cos_ang > 1 ? 1 : cos_ang);
            double sin_ang = std::sqrt(1.0 - cos_ang * cos_ang);
            if (std::fabs(sin_ang) <= (double)1e-3f) continue;
            double dir[3] = { (m[7] - m[5]) * 0.5 / sin_ang, (m[2] - m[6]) * 0.5 / sin_ang, (m[3] - m[1]) * 0.5 / sin_ang };
            double len = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            for (double& d : dir) d /= len;
            double ang = std::acos(cos_ang);
            ang += dis(gen) * noise_R_hi;
            double dw1 = dis(gen) * noise_R_hi, dw2 = dis(gen) * noise_R_hi, dw3 = dis(gen) * noise_R_hi;
            dir[0] += dw1; dir[1] += dw2; dir[2] += dw3;
            len = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            for (double& d : dir) d /= len;
            double s = std::sin(ang), c = std::cos(ang);
            double Kx[9] = { 0, -dir[2], dir[1], dir[2], 0, -dir[0], -dir[1], dir[0], 0 };
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double kk = Kx[3 * a] * Kx[b] + Kx[3 * a + 1] * Kx[3 + b] + Kx[3 * a + 2] * Kx[6 + b];
                    rt.R[(size_t)(3 * a + b)] = (a == b ? 1.0 : 0.0) + s * Kx[3 * a + b] + (1 - c) * kk;
                }
        }
    }
    BundleAdjustmentKanatani ba;
    BundleAdjustmentKanataniTermCriteria term_crit;
    if (allowed_repr_err > 0) term_crit.AllowedReprojErrRelativeChange(allowed_repr_err); // :287-288
    if (!dump_before.empty() && !DumpScene(dump_before, f0, map_noise, cams, track_rep, nullptr, &Ks)) return 3;
    std::fprintf(stderr, "start bundle adjustment...\n");
    bool op = ba.ComputeInplace(f0, map_noise, cams, track_rep, nullptr, &Ks, term_crit, max_iterations); // :291
    std::fprintf(stderr, "bundle adjustment finished with result: %d (%s)\n", (int)op, ba.OptimizationStatusString().c_str());
    if (!dump_after.empty() && !DumpScene(dump_after, f0, map_noise, cams, track_rep, nullptr, &Ks)) return 3;
    const srk_ba_report& r = ba.Report();
    std::printf("{\"result\": %d, \"status\": \"%s\", \"iterations\": %lld, \"attempts\": %lld, \"err_initial\": %.17g, "
                "\"err_final\": %.17g, \"frames\": %zu, \"points\": %zu}\n",
                (int)op, ba.OptimizationStatusString().c_str(), (long long)r.iterations, (long long)r.attempts, r.err_initial,
                r.err_final, M, noisy.size());
    return 0;
}
