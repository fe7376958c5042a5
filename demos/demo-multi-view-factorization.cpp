// demo-multi-view-factorization -- drop-in of the reference demo (cpp_impl/demos/demo-multi-view-factorization.cpp:351-661)
// and of the driver it runs, MultiViewIterativeFactorizer::IntegrateNewFrameCorners
// (cpp_impl/suriko-engine/src/multi-view-factorization.cpp:255-397), on the MI355X core: same flags (:351-370), same
// synthetic world (grid on a cosine surface with N(0, noise_x3D_std) jitter, mt19937 seeded 1234, :394-430), same camera
// path (:52-98) with the same rotation noise (:447-468), same pixel K, same frame loop:
//   frames 0, 1   ground-truth poses and points (:529-603, well_known_frames_count = 2);
//   frame f >= 2  corners from the ground-truth projection (DemoCornersMatcher, :302-348); anchor frame = the earlier
//                 frame sharing most reconstructed tracks (:39-60); depths of the common points in the anchor (:62-76);
//                 camera motion anchor -> f  = srk_mvf_relative_motion (:107-189); new landmarks = srk_mvf_estimate_depths
//                 (:223-253, :316-370); score = the driver's own ReprojError (:415-475); when it exceeds 1e-3: bundle
//                 adjustment in the driver's call contract -- shared K, f0 = 1, threshold 1e-3 (:379-394).
// No OpenCV window, no Pangolin thread, no glog (LOG lines go to stderr); the map mutex has no second thread to guard.
// Harness additions: --max_frames, --ba_max_iterations (the reference has no cap), --dump_ba_prefix=<path> (the scene of
// every BA call before / after, for the oracle comparison in tests/).  The last stdout line is a JSON summary.
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "flags.hpp"
#include "scene_dump.hpp"
#include "suriko_amd/bundle-adj-kanatani.hpp"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
using namespace suriko_amd;

namespace {
using V3 = std::array<double, 3>;

V3 sub(const V3& a, const V3& b) { return { a[0] - b[0], a[1] - b[1], a[2] - b[2] }; }
V3 add(const V3& a, const V3& b) { return { a[0] + b[0], a[1] + b[1], a[2] + b[2] }; }
V3 mul(const V3& a, double s) { return { a[0] * s, a[1] * s, a[2] * s }; }
double dot(const V3& a, const V3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
V3 cross(const V3& a, const V3& b) { return { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] }; }
V3 normalized(const V3& a) { return mul(a, 1 / std::sqrt(dot(a, a))); }
V3 matvec(const Matrix3& m, const V3& v) { return { m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2], m[6] * v[0] + m[7] * v[1] + m[8] * v[2] }; }
Matrix3 matmul(const Matrix3& a, const Matrix3& b)
{
    Matrix3 c{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[(size_t)(3 * i + j)] = a[(size_t)(3 * i)] * b[(size_t)j] + a[(size_t)(3 * i + 1)] * b[(size_t)(3 + j)] + a[(size_t)(3 * i + 2)] * b[(size_t)(6 + j)];
    return c;
}
Matrix3 transposed(const Matrix3& a) { return { a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8] }; }
V3 pt(const Point3& p) { return { p.x, p.y, p.z }; }
SE3Transform se3_inv(const SE3Transform& rt) // obs-geom.cpp:117-122
{
    SE3Transform r;
    r.R = transposed(rt.R);
    V3 t = matvec(r.R, pt(rt.T));
    r.T = { -t[0], -t[1], -t[2] };
    return r;
}
V3 se3_apply(const SE3Transform& rt, const V3& x) { return add(matvec(rt.R, x), pt(rt.T)); } // :124-127
SE3Transform se3_compose(const SE3Transform& a, const SE3Transform& b) // a o b, :129-139
{
    SE3Transform r;
    r.R = matmul(a.R, b.R);
    V3 t = add(matvec(a.R, pt(b.T)), pt(a.T));
    r.T = { t[0], t[1], t[2] };
    return r;
}
// LookAtLufWfc (obs-geom.cpp:729-749): camera-to-world with X = left, Y = up, Z = forward
SE3Transform look_at_luf_wfc(const V3& eye, const V3& center, const V3& up)
{
    V3 fwd = normalized(sub(center, eye));
    V3 cup = normalized(sub(up, mul(fwd, dot(up, fwd))));
    V3 left = cross(cup, fwd);
    SE3Transform w;
    w.R = { left[0], cup[0], fwd[0], left[1], cup[1], fwd[1], left[2], cup[2], fwd[2] };
    w.T = { eye[0], eye[1], eye[2] };
    return w;
}
// AxisAngleFromRotMat = LogSO3 x angle (obs-geom.cpp:562-604): false when sin(angle) ~ 0
bool axis_angle_from_rot_mat(const Matrix3& R, double w[3])
{
    double cos_ang = 0.5 * (R[0] + R[4] + R[8] - 1);
    cos_ang = cos_ang < -1 ? -1 : (cos_ang > 1 ? 1 : cos_ang);
    const double sin_ang = std::sqrt(1.0 - cos_ang * cos_ang);
    if (std::fabs(sin_ang) <= (double)1e-3f) return false;
    V3 d = { R[7] - R[5], R[2] - R[6], R[3] - R[1] };
    d = normalized(mul(d, 0.5 / sin_ang));
    const double ang = std::acos(cos_ang);
    for (int i = 0; i < 3; ++i) w[i] = d[(size_t)i] * ang;
    return true;
}
// RotMatFromAxisAngle (obs-geom.cpp:520-561, Rodrigues): false for a zero vector
bool rot_mat_from_axis_angle(const double w[3], Matrix3* R)
{
    const double ang = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (std::fabs(ang) <= 1e-8) return false; // IsClose(0, ang): atol 1e-8
    const double d[3] = { w[0] / ang, w[1] / ang, w[2] / ang }, s = std::sin(ang), c = std::cos(ang);
    const Matrix3 Kx = { 0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0 };
    const Matrix3 KK = matmul(Kx, Kx);
    for (int i = 0; i < 9; ++i) (*R)[(size_t)i] = ((i % 4 == 0) ? 1.0 : 0.0) + s * Kx[(size_t)i] + (1 - c) * KK[(size_t)i];
    return true;
}
// demo-multi-view-factorization.cpp:52-98
void camera_shots_along_rectangular_path(double xmin, double xmax, double ymin, double ymax, double zmin, size_t steps_x,
                                         size_t steps_y, double down_offset, double ascent_z, std::vector<SE3Transform>* cams)
{
    const std::array<V3, 5> base = { V3{ xmax, ymin, zmin }, V3{ xmin, ymin, zmin }, V3{ xmin, ymax, zmin }, V3{ xmax, ymax, zmin }, V3{ xmax, ymin, zmin } };
    const std::array<size_t, 4> steps = { steps_x, steps_y, steps_x, steps_y };
    const double skew = std::atan2(std::fabs(xmax - xmin), std::fabs(ymax - ymin));
    const double offx = down_offset * std::sin(skew), offy = -down_offset * std::cos(skew);
    const Matrix3 rotz_pi = { std::cos(M_PI), -std::sin(M_PI), 0, std::sin(M_PI), std::cos(M_PI), 0, 0, 0, 1 }; // RotMat(0, 0, 1, pi)
    for (size_t b = 0; b + 1 < base.size(); ++b) {
        const V3 step = mul(sub(base[b + 1], base[b]), 1.0 / (double)steps[b]);
        for (size_t k = 0; k < steps[b]; ++k) {
            const V3 cur = add(base[b], mul(step, (double)k));
            SE3Transform wfc = look_at_luf_wfc(add(cur, V3{ offx, offy, ascent_z }), cur, V3{ 0, 0, 1 });
            wfc.R = matmul(wfc.R, rotz_pi); // left-up-forward -> right-down-forward
            cams->push_back(se3_inv(wfc));
        }
    }
}
} // namespace

int main(int argc, char** argv)
{
    Flags fl;
    fl.Parse(argc, argv);
    const double xmin = fl.Double("world_xmin", -1), xmax = fl.Double("world_xmax", 1);
    const double ymin = fl.Double("world_ymin", -1), ymax = fl.Double("world_ymax", 1);
    const double zmin = fl.Double("world_zmin", 0), zmax = fl.Double("world_zmax", 1);
    const double cell_x = fl.Double("world_cell_size_x", 0.5), cell_y = fl.Double("world_cell_size_y", 0.5);
    const double down_offset = fl.Double("viewer_offset_down", 10), ascent_z = fl.Double("viewer_ascendZ", 10);
    const long steps_x = fl.Int("viewer_steps_per_side_x", 10), steps_y = fl.Int("viewer_steps_per_side_y", 10);
    const double noise_R_std = fl.Double("noise_R_std", 0.005), noise_x3D_std = fl.Double("noise_x3D_std", 0.005);
    const bool skim_over = fl.Bool("debug_skim_over", true);
    const bool fake_mapping = fl.Bool("fake_mapping", false), fake_localization = fl.Bool("fake_localization", false);
    const long max_frames = fl.Int("max_frames", 0), ba_max_iterations = fl.Int("ba_max_iterations", 0);
    const std::string dump_prefix = fl.String("dump_ba_prefix", "");
    std::fprintf(stderr, "noise_x3D_std=%g\nnoise_R_std=%g\n", noise_x3D_std, noise_R_std);

    // ---- the synthetic world (:394-430)
    const double gap = 1e-8;
    std::mt19937 gen(1234);
    std::normal_distribution<double> x3d_noise(0, noise_x3D_std > 0 ? noise_x3D_std : 1.0);
    std::vector<V3> world;
    const double xmid = (xmin + xmax) / 2, xlen = xmax - xmin, zlen = zmax - zmin;
    for (double gx = xmin; gx < xmax + gap; gx += cell_x)
        for (double gy = ymin; gy < ymax + gap; gy += cell_y) {
            double x = gx, y = gy, z = zmin + std::cos((x - xmid) / xlen * M_PI) * zlen;
            if (noise_x3D_std > 0) { x += x3d_noise(gen); y += x3d_noise(gen); z += x3d_noise(gen); }
            world.push_back({ x, y, z });
        }
    std::fprintf(stderr, "points_count=%zu\n", world.size());
    const std::array<double, 2> img = { 800, 600 };
    const Matrix3 K = { 880, 0, img[0] / 2, 0, 660, img[1] / 2, 0, 0, 1 };
    const Matrix3 Kinv = { 1 / K[0], 0, -K[2] / K[0], 0, 1 / K[4], -K[5] / K[4], 0, 0, 1 };
    std::vector<SE3Transform> gt;
    camera_shots_along_rectangular_path(xmin, xmax, ymin, ymax, zmin, (size_t)steps_x, (size_t)steps_y, down_offset, ascent_z, &gt);
    if (noise_R_std > 0) { // :447-468 (AxisAngleFromRotMat / RotMatFromAxisAngle through the library's Rodrigues pair)
        std::normal_distribution<double> rn(0, noise_R_std);
        for (SE3Transform& c : gt) {
            double w[3];
            if (axis_angle_from_rot_mat(c.R, w)) {
                const double d1 = rn(gen), d2 = rn(gen), d3 = rn(gen);
                w[0] += d1; w[1] += d2; w[2] += d3;
                Matrix3 Rn;
                if (rot_mat_from_axis_angle(w, &Rn)) c.R = Rn;
            }
        }
    }
    size_t frames_count = gt.size();
    if (max_frames > 0 && (size_t)max_frames < frames_count) frames_count = (size_t)max_frames;
    std::fprintf(stderr, "frames_count=%zu\n", frames_count);

    // ---- the factorizer's state (multi-view-factorization.h:17-60)
    FragmentMap map(2000000);
    CornerTrackRepository tracks;
    std::vector<SE3Transform> cams; // cam_orient_cfw_
    std::vector<long> track_of_world(world.size(), -1); // SyntheticVirtualPointId -> track
    BundleAdjustmentKanatani ba;
    const double kF0 = 1;
    long ba_calls = 0, ba_iterations = 0, ba_attempts = 0, integrated = 0, failed = 0;
    double last_err = -1, max_pose_diff = 0, ba_ms_total = 0, ba_ms_max = 0;
    size_t ba_last_points = 0, ba_last_frames = 0, ba_last_seen = 0;

    auto image_coord = [&](const Point2f& pix) { return matvec(Kinv, V3{ pix.x, pix.y, 1 }); };
    auto detect_and_match = [&](size_t f, bool create_points) { // DemoCornersMatcher (:302-348) / the demo's own loop (:548-600)
        for (size_t i = 0; i < world.size(); ++i) {
            const V3 h = matvec(K, se3_apply(gt[f], world[i]));
            const double px = h[0] / h[2], py = h[1] / h[2];
            if (!(px >= 0 && px < img[0] && py >= 0 && py < img[1])) continue;
            long t = track_of_world[i];
            if (t < 0) {
                CornerTrack& nt = tracks.AddCornerTrackObj();
                if (create_points) nt.SalientPointId = map.AddSalientPoint({ world[i][0], world[i][1], world[i][2] });
                t = track_of_world[i] = (long)nt.TrackId;
            }
            tracks.CornerTracks[(size_t)t].AddCorner(f, { px, py });
        }
    };
    auto mvf_error = [&](double* err) { return ba.ReprojErrorMvf(kF0, map, cams, tracks, &K, err); };

    const size_t well_known = 2;
    for (size_t f = 0; f < frames_count; ++f) {
        if (skim_over || f < well_known) {
            cams.push_back(gt[f]);
            detect_and_match(f, true);
            if (f < well_known) { double e = -1; bool op = mvf_error(&e); std::fprintf(stderr, "ReprojError=%g\n", op ? e : -1.0); }
            continue;
        }
        // ---- IntegrateNewFrameCorners (:255-397)
        detect_and_match(f, false);
        std::vector<size_t> in_frame; // tracks with a corner in the new frame (PopulateCornerTrackIds)
        for (const CornerTrack& t : tracks.CornerTracks)
            if (t.GetCorner(f)) in_frame.push_back(t.TrackId);
        // FindAnchorFrame (:39-60): the first earlier frame with the most common reconstructed tracks
        size_t anchor = 0, best = 0;
        for (size_t p = 0; p < f; ++p) {
            size_t cnt = 0;
            for (size_t id : in_frame)
                if (tracks.CornerTracks[id].GetCorner(p) && tracks.CornerTracks[id].SalientPointId) ++cnt;
            if (cnt > best) { best = cnt; anchor = p; }
        }
        // The reference gives up here as well: IntegrateNewFrameCorners returns false before it adds a pose, its next call
        // re-detects the same frame index (FramesCount()) and fails again (:262-270, "TODO: how to roll back").  The
        // drop-in stops integrating instead of repeating the failure for every remaining frame.
        if (best == 0) { std::fprintf(stderr, "Can't integrate frameInd=%zu\n", f); failed = (long)(frames_count - f); break; }
        std::vector<double> xa, xt, depth;
        for (size_t id : in_frame) {
            const CornerTrack& t = tracks.CornerTracks[id];
            if (!t.GetCorner(anchor) || !t.SalientPointId) continue;
            const V3 a = image_coord(*t.GetCorner(anchor)), b = image_coord(*t.GetCorner(f));
            xa.insert(xa.end(), a.begin(), a.end());
            xt.insert(xt.end(), b.begin(), b.end());
            depth.push_back(se3_apply(cams[anchor], pt(map.GetSalientPoint(*t.SalientPointId)))[2]); // Get3DPointDepth :62-76
        }
        std::fprintf(stderr, "f=%zu anchored on f=%zu using common_points=%zu\n", f, anchor, depth.size());
        SE3Transform new_from_anchor;
        double Tn[3];
        const int mo = srk_mvf_relative_motion(ba.Handle(), (int64_t)depth.size(), xa.data(), xt.data(), depth.data(), new_from_anchor.R.data(), Tn);
        // fewer than 6 common points leave the 12-unknown system rank deficient (the reference takes whatever vector its
        // SVD returns and carries on with a meaningless pose); srk_mvf_relative_motion refuses, and tracking ends here
        if (mo != 1) {
            std::fprintf(stderr, "tracking lost at frame %zu: relative motion from %zu common points failed (%d)\n", f, depth.size(), mo);
            failed = (long)(frames_count - f);
            break;
        }
        new_from_anchor.T = { Tn[0], Tn[1], Tn[2] };
        { // distance from the ground-truth motion, as the reference logs it (:283-295)
            const SE3Transform gtm = se3_compose(gt[f], se3_inv(gt[anchor]));
            double dd = 0;
            for (int e = 0; e < 9; ++e) dd += (gtm.R[(size_t)e] - new_from_anchor.R[(size_t)e]) * (gtm.R[(size_t)e] - new_from_anchor.R[(size_t)e]);
            const V3 dt = sub(pt(gtm.T), pt(new_from_anchor.T));
            const double diff = std::sqrt(dd) + std::sqrt(dot(dt, dt));
            if (diff > max_pose_diff) max_pose_diff = diff;
            std::fprintf(stderr, "%scam localiz, diff_value=%g frame_ind=%zu\n", diff > 1 ? "diverged " : "", diff, f);
        }
        cams.push_back(fake_localization ? gt[f] : se3_compose(new_from_anchor, cams[anchor]));
        // new landmarks: tracks seen in this frame, not yet reconstructed, with at least two corners (:316-370)
        std::vector<size_t> cand;
        std::vector<int64_t> rp{ 0 };
        std::vector<int32_t> fr;
        std::vector<double> xm;
        for (size_t id : in_frame) {
            const CornerTrack& t = tracks.CornerTracks[id];
            if (t.SalientPointId) continue;
            size_t n = 0;
            for (size_t p = 0; p <= f; ++p)
                if (auto c = t.GetCorner(p)) { const V3 m = image_coord(*c); fr.push_back((int32_t)p); xm.insert(xm.end(), m.begin(), m.end()); ++n; }
            if (n <= 1) { fr.resize(fr.size() - n); xm.resize(xm.size() - 3 * n); continue; }
            cand.push_back(id);
            rp.push_back((int64_t)fr.size());
        }
        size_t reconstructed = 0;
        if (!cand.empty()) {
            std::vector<double> R(9 * cams.size()), T(3 * cams.size()), dep(cand.size());
            for (size_t j = 0; j < cams.size(); ++j) {
                for (int e = 0; e < 9; ++e) R[9 * j + (size_t)e] = cams[j].R[(size_t)e];
                T[3 * j] = cams[j].T.x; T[3 * j + 1] = cams[j].T.y; T[3 * j + 2] = cams[j].T.z;
            }
            if (srk_mvf_estimate_depths(ba.Handle(), (int64_t)cand.size(), rp.data(), fr.data(), xm.data(), (int32_t)cams.size(), R.data(), T.data(), dep.data()) != 0) return 4;
            for (size_t k = 0; k < cand.size(); ++k) {
                CornerTrack& t = tracks.CornerTracks[cand[k]];
                const size_t base = (size_t)fr[(size_t)rp[k]];
                V3 xw = se3_apply(se3_inv(cams[base]), mul(V3{ xm[3 * (size_t)rp[k]], xm[3 * (size_t)rp[k] + 1], xm[3 * (size_t)rp[k] + 2] }, dep[k]));
                if (fake_mapping)
                    for (size_t i = 0; i < world.size(); ++i)
                        if (track_of_world[i] == (long)t.TrackId) xw = world[i];
                t.SalientPointId = map.AddSalientPoint({ xw[0], xw[1], xw[2] });
                ++reconstructed;
            }
        }
        double err = -1;
        bool op = mvf_error(&err);
        std::fprintf(stderr, "f=%zu reconstructed_salient_points_count=%zu ReprojError=%g\n", f, reconstructed, op ? err : -1.0);
        if (op && err > (double)1e-3f) { // :377-394
            std::fprintf(stderr, "start bundle adjustment...\n");
            BundleAdjustmentKanataniTermCriteria crit;
            crit.AllowedReprojErrRelativeChange(1e-3);
            if (!dump_prefix.empty()) DumpScene(dump_prefix + "_" + std::to_string(ba_calls) + "_before.bin", kF0, map, cams, tracks, &K, nullptr);
            const auto t_ba = std::chrono::steady_clock::now();
            op = ba.ComputeInplace(kF0, map, cams, tracks, &K, nullptr, crit, ba_max_iterations);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ba).count();
            ba_ms_total += ms;
            if (ms > ba_ms_max) ba_ms_max = ms;
            ba_last_points = ba.PointsCount();
            ba_last_frames = ba.FramesCount();
            ba_last_seen = (size_t)ba.Report().seen;
            if (!dump_prefix.empty()) DumpScene(dump_prefix + "_" + std::to_string(ba_calls) + "_after.bin", kF0, map, cams, tracks, &K, nullptr);
            std::fprintf(stderr, "bundle adjustment finished with result: %d (%s)\n", (int)op, ba.OptimizationStatusString().c_str());
            ++ba_calls;
            ba_iterations += (long)ba.Report().iterations;
            ba_attempts += (long)ba.Report().attempts;
            if (mvf_error(&err)) std::fprintf(stderr, "ReprojError=%g\n", err);
        }
        last_err = err;
        ++integrated;
    }
    size_t reconstructed_points = 0;
    for (const CornerTrack& t : tracks.CornerTracks) reconstructed_points += t.SalientPointId ? 1 : 0;
    std::printf("{\"frames\": %zu, \"world_points\": %zu, \"tracks\": %zu, \"salient_points\": %zu, \"integrated_frames\": %ld, "
                "\"failed_frames\": %ld, \"ba_calls\": %ld, \"ba_iterations\": %ld, \"ba_attempts\": %ld, \"last_reproj_err\": %.17g, "
                "\"max_pose_diff\": %.6g, \"ba_ms_total\": %.3f, \"ba_ms_max\": %.3f, \"ba_last_points\": %zu, "
                "\"ba_last_frames\": %zu, \"ba_last_observations\": %zu}\n",
                cams.size(), world.size(), tracks.CornerTracks.size(), reconstructed_points, integrated, failed, ba_calls,
                ba_iterations, ba_attempts, last_err, max_pose_diff, ba_ms_total, ba_ms_max, ba_last_points, ba_last_frames,
                ba_last_seen);
    return 0;
}
