// bundle-adj-kanatani.hpp -- C++17 adapter with the class API of whigg/surikatoko's BundleAdjustmentKanatani
// (cpp_impl/suriko-engine/include/suriko/bundle-adj-kanatani.h:96-261) on top of the C ABI of libsrk_ba.so
// (include/srk_ba.h).  Header-only; no Eigen, glog or GSL (not available in this image): the containers below
// carry exactly the data the reference's FragmentMap / CornerTrackRepository / SE3Transform hand to the BA call.
//
// A maintainer of the reference swaps the body of suriko::BundleAdjustmentKanatani::ComputeInplace for a call to
// suriko_amd::BundleAdjustmentKanatani::ComputeInplace after copying its Eigen containers into these plain ones
// (INTEGRATION.md shows the ~40-line shim).  Semantics kept: in-place update of points and inverse camera poses,
// exactly one of shared_K / Ks, `bool` result + OptimizationStatusString(), K never modified (reference quirk,
// bundle-adj-kanatani.cpp:2027-2034), pnt_ind = order of tracks that have a SalientPointId (:1161-1169) while
// coordinates are fetched by salient-point id (:1171).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../srk_ba.h"

namespace suriko_amd {

using Scalar = double; // rt-config.h:41-48 (default)

struct Point3 { Scalar x = 0, y = 0, z = 0; };
struct Point2f { Scalar x = 0, y = 0; };
using Matrix3 = std::array<Scalar, 9>; // row-major

/// obs-geom.h:177-190.  Inverse orientation: world -> camera.
struct SE3Transform {
    Point3 T;
    Matrix3 R{ 1, 0, 0, 0, 1, 0, 0, 0, 1 };
};

/// obs-geom.h:207-243: id -> index = id - offset - 1 (obs-geom.cpp:247-251), default offset 1'000'000.
class FragmentMap {
public:
    explicit FragmentMap(size_t fragment_id_offset = 1000000) : offset_(fragment_id_offset) {}
    size_t AddSalientPoint(const Point3& p) { pts_.push_back(p); return pts_.size() + offset_; }
    size_t SalientPointsCount() const { return pts_.size(); }
    Point3& GetSalientPoint(size_t id) { return pts_.at(id - offset_ - 1); }
    const Point3& GetSalientPoint(size_t id) const { return pts_.at(id - offset_ - 1); }
private:
    std::vector<Point3> pts_;
    size_t offset_;
};

/// obs-geom.h:251-283: a start frame + one optional corner per following frame (gaps allowed).
struct CornerTrack {
    size_t TrackId = 0;
    std::optional<size_t> SalientPointId;
    ptrdiff_t StartFrameInd = -1;
    std::vector<std::optional<Point2f>> CoordPerFramePixels;
    void AddCorner(size_t frame_ind, const Point2f& v) {
        if (StartFrameInd == -1) StartFrameInd = (ptrdiff_t)frame_ind;
        if ((ptrdiff_t)frame_ind < StartFrameInd) throw std::invalid_argument("corner before the start frame");
        CoordPerFramePixels.resize(frame_ind - (size_t)StartFrameInd + 1);
        CoordPerFramePixels.back() = v;
    }
    std::optional<Point2f> GetCorner(size_t frame_ind) const {
        ptrdiff_t k = (ptrdiff_t)frame_ind - StartFrameInd;
        if (StartFrameInd == -1 || k < 0 || (size_t)k >= CoordPerFramePixels.size()) return std::nullopt;
        return CoordPerFramePixels[(size_t)k];
    }
};
struct CornerTrackRepository {
    std::vector<CornerTrack> CornerTracks;
    CornerTrack& AddCornerTrackObj() { CornerTracks.emplace_back(); CornerTracks.back().TrackId = CornerTracks.size() - 1; return CornerTracks.back(); }
};

/// bundle-adj-kanatani.h:68-92
class BundleAdjustmentKanataniTermCriteria {
public:
    void AllowedReprojErrRelativeChange(std::optional<Scalar> v) { rel_ = v; }
    std::optional<Scalar> AllowedReprojErrRelativeChange() const { return rel_; }
    void MaxHessianFactor(std::optional<Scalar> v) { maxf_ = v; }
    std::optional<Scalar> MaxHessianFactor() const { return maxf_; }
private:
    std::optional<Scalar> rel_, maxf_;
};

class BundleAdjustmentKanatani {
public:
    explicit BundleAdjustmentKanatani(int device_id = 0) : h_(srk_ba_create(device_id)) {
        if (!h_) throw std::runtime_error("srk_ba_create failed: no usable HIP device (there is no CPU fallback)");
    }
    ~BundleAdjustmentKanatani() { srk_ba_destroy(h_); }
    BundleAdjustmentKanatani(const BundleAdjustmentKanatani&) = delete;
    BundleAdjustmentKanatani& operator=(const BundleAdjustmentKanatani&) = delete;

    /// bundle-adj-kanatani.h:179-184.  Throws std::invalid_argument where the reference CHECK-aborts
    /// (f0 ~ 0, both or neither K given, :420-421).
    bool ComputeInplace(Scalar f0, FragmentMap& map, std::vector<SE3Transform>& inverse_orient_cams,
                        const CornerTrackRepository& track_rep, const Matrix3* shared_intrinsic_cam_mat,
                        std::vector<Matrix3>* intrinsic_cam_mats, const BundleAdjustmentKanataniTermCriteria& term_crit,
                        int64_t max_iterations = 0) {
        Flat f = Flatten(f0, map, inverse_orient_cams, track_rep, shared_intrinsic_cam_mat, intrinsic_cam_mats);
        Scalar a = term_crit.AllowedReprojErrRelativeChange().value_or(0), m = term_crit.MaxHessianFactor().value_or(0);
        int rc = srk_ba_compute_inplace(h_, f0, (int64_t)f.ids.size(), f.pts.data(), (int32_t)inverse_orient_cams.size(),
                                        f.R.data(), f.T.data(), f.K.data(), f.shared, f.row_ptr.data(), f.frames.data(),
                                        f.uv.data(), term_crit.AllowedReprojErrRelativeChange() ? &a : nullptr,
                                        term_crit.MaxHessianFactor() ? &m : nullptr, max_iterations, &report_);
        if (rc < 0) Raise(rc);
        status_ = srk_ba_status_string(report_.status);
        f0_ = f0;
        points_count_ = f.ids.size();
        frames_count_ = inverse_orient_cams.size();
        // write back: points by salient-point id, cameras in place (K is never modified)
        for (size_t i = 0; i < f.ids.size(); ++i) {
            Point3& p = map.GetSalientPoint(f.ids[i]);
            p.x = f.pts[3 * i]; p.y = f.pts[3 * i + 1]; p.z = f.pts[3 * i + 2];
        }
        for (size_t j = 0; j < inverse_orient_cams.size(); ++j) {
            for (int e = 0; e < 9; ++e) inverse_orient_cams[j].R[(size_t)e] = f.R[9 * j + (size_t)e];
            inverse_orient_cams[j].T = { f.T[3 * j], f.T[3 * j + 1], f.T[3 * j + 2] };
        }
        return rc == 0;
    }

    /// bundle-adj-kanatani.h:167-172 (static in the reference; needs a device here)
    Scalar ReprojError(Scalar f0, const FragmentMap& map, const std::vector<SE3Transform>& inverse_orient_cams,
                       const CornerTrackRepository& track_rep, const Matrix3* shared_intrinsic_cam_mat = nullptr,
                       const std::vector<Matrix3>* intrinsic_cam_mats = nullptr, size_t* seen_points_count = nullptr) {
        Flat f = Flatten(f0, map, inverse_orient_cams, track_rep, shared_intrinsic_cam_mat, intrinsic_cam_mats);
        int64_t seen = 0;
        Scalar e = srk_ba_reproj_error(h_, f0, (int64_t)f.ids.size(), f.pts.data(), (int32_t)inverse_orient_cams.size(),
                                       f.R.data(), f.T.data(), f.K.data(), f.shared, f.row_ptr.data(), f.frames.data(),
                                       f.uv.data(), &seen);
        if (std::isnan(e)) throw std::runtime_error(std::string("srk_ba_reproj_error: ") + srk_ba_last_error(h_));
        if (seen_points_count) *seen_points_count = (size_t)seen;
        f0_ = f0;
        return e;
    }
    /// MultiViewIterativeFactorizer::ReprojError (multi-view-factorization.cpp:415-475): the score the MVF driver takes
    /// before deciding to run BA; shared K only, observations with |z| <= 1e-5 skipped, false when nothing was summed.
    bool ReprojErrorMvf(Scalar f0, const FragmentMap& map, const std::vector<SE3Transform>& cam_orient_cfw,
                        const CornerTrackRepository& track_rep, const Matrix3* shared_intrinsic_cam_mat, Scalar* reproj_err) {
        Flat f = Flatten(f0, map, cam_orient_cfw, track_rep, shared_intrinsic_cam_mat, (const std::vector<Matrix3>*)nullptr);
        int64_t n = 0;
        int rc = srk_ba_reproj_error_mvf(h_, f0, (int64_t)f.ids.size(), f.pts.data(), (int32_t)cam_orient_cfw.size(),
                                         f.R.data(), f.T.data(), f.K.data(), f.shared, f.row_ptr.data(), f.frames.data(),
                                         f.uv.data(), 1e-5, reproj_err, &n);
        if (rc < 0) Raise(rc);
        return rc == 1;
    }
    Scalar ReprojErrorPixPerPoint(Scalar reproj_err, size_t seen) const { return f0_ * std::sqrt(reproj_err / (Scalar)seen); } // .cpp:602-615

    size_t PointsCount() const { return points_count_; }
    size_t FramesCount() const { return frames_count_; }
    size_t VarsCount() const { return 3 * points_count_ + 10 * frames_count_; }
    size_t NormalizedVarsCount() const { return VarsCount() - 7; }
    const std::string& OptimizationStatusString() const { return status_; }
    const srk_ba_report& Report() const { return report_; }
    /// the C-ABI handle, for callers that also use the srk_mvf_* steps on the same device (the MVF driver)
    srk_ba* Handle() const { return h_; }

private:
    struct Flat {
        std::vector<size_t> ids;
        std::vector<Scalar> pts, R, T, K, uv;
        std::vector<int64_t> row_ptr;
        std::vector<int32_t> frames;
        int shared = 0;
    };
    template <class KVec>
    Flat Flatten(Scalar f0, const FragmentMap& map, const std::vector<SE3Transform>& cams, const CornerTrackRepository& tr,
                 const Matrix3* shared_K, const KVec* Ks) const {
        if (std::fabs(f0) <= 1e-8) throw std::invalid_argument("f0 != 0");                      // .cpp:420
        if ((shared_K != nullptr) == (Ks != nullptr)) throw std::invalid_argument("Provide either shared K or separate K for each camera frame"); // :421
        Flat f;
        f.row_ptr.push_back(0);
        for (const CornerTrack& t : tr.CornerTracks) { // pnt_ind order (:1161-1169)
            if (!t.SalientPointId) continue;
            const Point3& p = map.GetSalientPoint(*t.SalientPointId);
            f.ids.push_back(*t.SalientPointId);
            f.pts.insert(f.pts.end(), { p.x, p.y, p.z });
            for (size_t j = 0; j < cams.size(); ++j)
                if (auto c = t.GetCorner(j)) { f.frames.push_back((int32_t)j); f.uv.push_back(c->x); f.uv.push_back(c->y); }
            f.row_ptr.push_back((int64_t)f.frames.size());
        }
        for (const SE3Transform& c : cams) {
            f.R.insert(f.R.end(), c.R.begin(), c.R.end());
            f.T.insert(f.T.end(), { c.T.x, c.T.y, c.T.z });
        }
        if (shared_K) { f.K.assign(shared_K->begin(), shared_K->end()); f.shared = 1; }
        else for (const Matrix3& k : *Ks) f.K.insert(f.K.end(), k.begin(), k.end());
        return f;
    }
    [[noreturn]] void Raise(int rc) const {
        std::string msg = srk_ba_last_error(h_);
        if (rc == SRK_E_ARGS) throw std::invalid_argument(msg);
        throw std::runtime_error("srk_ba: " + msg);
    }
    srk_ba* h_;
    srk_ba_report report_{};
    std::string status_;
    Scalar f0_ = 0;
    size_t points_count_ = 0, frames_count_ = 0;
};

} // namespace suriko_amd
