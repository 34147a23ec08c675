// TEST DOUBLE -- the declarations of the reference's classes that shim/*.cc touches, reduced to those members (same
// names and types as leavesnight/VIEO_SLAM: include/FrameBase.h, Frame.h, KeyFrame.h, MapPoint.h, Map.h, ORBmatcher.h,
// Optimizer.h, FrameBase_impl.h, src/Odom/NavState.h, OdomPreIntegrator.h, OdomData.h, common/camera_models/
// camera_base.h).  Lets `g++ -fsyntax-only` type-check the shims where the real tree cannot be compiled (no OpenCV /
// Eigen / Sophus in the image).  Declarations only; nothing is linked or run; no parity evidence.
#pragma once
#include <array>
#include <cmath>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include "third_party_decls.hpp"
#include "vieo_shim.hpp"  // VIEO_SLAM::ORBextractor: include/vieo_shim.hpp REPLACES the reference's include/ORBextractor.h

namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
}  // namespace DBoW2

namespace Eigen {
template <class T>
using aligned_vector = std::vector<T, Eigen::aligned_allocator<T>>;  // common/eigen_utils.h:20
template <class T>
using aligned_list = std::list<T, Eigen::aligned_allocator<T>>;
}  // namespace Eigen

namespace VIEO_SLAM {
using std::set;
using std::vector;
using Eigen::aligned_list;
#define listeig(EncData) Eigen::aligned_list<EncData>
using Eigen::aligned_vector;
using Eigen::Vector3d;
class ORBVocabulary;
using Eigen::Matrix;
typedef Eigen::Matrix<double, 6, 1> Vector6d;
typedef Eigen::Matrix<double, 6, 6> Matrix6d;
typedef Eigen::Matrix<double, 9, 9> Matrix9d;

class NavState {
 public:
  Sophus::SO3exd mRwb;
  Eigen::Vector3d mpwb, mvwb, mbg, mba, mdbg, mdba;
};

class IMUDataBase {
 public:
  static Eigen::Matrix3d mSigmag, mSigmaa;
  static double mInvSigmabg2, mInvSigmaba2;
  static int mdt_cov_noise_fixed;
  static double mFreqRef;
  double mtm;
  Eigen::Vector3d mw, ma;
};
typedef IMUDataBase IMUData;
class EncPreIntegrator {
 public:
  double mdeltatij;
  Vector6d mdelxEij;
  Matrix6d mSigmaEij;
};
// src/Odom/OdomPreIntegrator.h, reduced: the pre-integrator is a class template over the sample type
template <class _OdomData>
class OdomPreIntegratorBase {
 public:
  aligned_list<_OdomData>& GetRawDataRef();
  const aligned_list<_OdomData>& GetRawDataRef() const;
  double mdeltatij;
};
template <class IMUDataBase>
class IMUPreIntegratorBase : public OdomPreIntegratorBase<IMUDataBase> {
 public:
  Eigen::Matrix3d mRij;
  Eigen::Vector3d mvij, mpij;
  Matrix9d mSigmaijPRV, mSigmaij;
  Eigen::Matrix3d mJgpij, mJapij, mJgvij, mJavij, mJgRij;
  int PreIntegration(const double& timeStampi, const double& timeStampj, const Eigen::Vector3d& bgi_bar, const Eigen::Vector3d& bai_bar,
                     const typename listeig(IMUDataBase)::const_iterator& iterBegin,
                     const typename listeig(IMUDataBase)::const_iterator& iterEnd, bool breset = true);
  void reset();
};
typedef IMUPreIntegratorBase<IMUDataBase> IMUPreintegrator;

namespace camm {
class GeometricCamera {
 public:
  typedef std::shared_ptr<GeometricCamera> Ptr;
  enum CameraModel { kUnknown = -1, kPinhole, kRadtan, kKB8 };
  const Sophus::SE3<float>& GetTrc() const;
  const Sophus::SE3<float>& GetTcr() const;
  const std::vector<float>& GetParameters() const;
  const CameraModel& camera_model() const;
  struct PairHash {
    size_t operator()(const std::pair<size_t, size_t>& p) const;
  };
  using MapCamIdx2Idx = std::unordered_map<std::pair<size_t, size_t>, size_t, PairHash>;
};
using Camera = GeometricCamera;
class KB8Camera : public GeometricCamera {
 public:
  const std::vector<int>& GetvLappingArea() const;
};
}  // namespace camm

class MapPoint;
class KeyFrame;
class Frame;
class Map;

class FrameBase {
 public:
  virtual ~FrameBase();
  virtual const Sophus::SE3d GetTwc();
  virtual const Sophus::SE3d GetTcw();
  virtual void AddMapPoint(MapPoint* pMP, const size_t& idx);
  virtual void EraseMapPointMatch(const size_t& idx);
  virtual vector<MapPoint*> GetMapPointMatches();
  virtual NavState GetNavState(void);
  virtual void SetNavState(const NavState& ns);
  virtual EncPreIntegrator GetEncPreInt(void);
  virtual IMUPreintegrator GetIMUPreInt(void);
  virtual bool isBad();
  void AssignFeaturesToGrid();
  void ComputeImageBounds(const vector<int>& wid_hei);
  double timestamp_, ftimestamp_;
  static cv::Mat mTbc, mTce;
  static Eigen::Matrix3d meigRcb;
  static Eigen::Vector3d meigtcb;
  static bool usedistort_;
  vector<camm::Camera::Ptr> mpCameras;
  int N;
  vector<cv::KeyPoint> mvKeys, mvKeysUn;
  vector<std::pair<size_t, size_t>> mapn2in_;
  unsigned long nid_;
  float mThDepth;
  cv::Mat mDescriptors;
  DBoW2::FeatureVector mFeatVec;
  struct StereoInfo {
    vector<float> vdepth_, vuright_;
    aligned_vector<Vector3d> v3dpoints_;
    vector<bool> goodmatches_;
    camm::Camera::MapCamIdx2Idx mapcamidx2idxs_;
    float baseline_bf_[2];
  } stereoinfo_;
  struct ScalePyramidInfo {
    vector<float> vscalefactor_;
    float fscalefactor_, flogscalefactor_;
    vector<float> vlevelsigma2_, vinvlevelsigma2_;
  } scalepyrinfo_;

 protected:
  const Sophus::SE3d GetTcwCst() const;
  struct GridInfo {
    vector<std::array<float, 4>> minmax_xy_;
  };
  static GridInfo gridinfo_;
  vector<MapPoint*> mvpMapPoints;
};

class Frame : public FrameBase {
 public:
  Frame();
  Frame(const vector<cv::Mat>& ims, const double& timeStamp, const vector<ORBextractor*>& extractors, ORBVocabulary* voc,
        const vector<camm::Camera::Ptr>& CamInsts, const float& bf, const float& thDepth,
        IMUPreintegrator* ppreint_imu_kf = nullptr, EncPreIntegrator* ppreint_enc_kf = nullptr, bool usedistort = true,
        const float th_far_pts = 0);
  void ComputeStereoMatches();
  void ComputeStereoFishEyeMatches(const float th_far_pts = 0);
  void SetPose(cv::Mat Tcw);
  vector<ORBextractor*> mpORBextractors;
  vector<size_t> num_mono;
  std::vector<std::vector<cv::KeyPoint>> vvkeys_;
  std::vector<cv::Mat> vdescriptors_;
  vector<vector<size_t>> mvidxsMatches;
  vector<size_t> mapidxs2n_;
  std::vector<std::vector<size_t>> mapin2n_;
  Matrix<double, 15, 15> mMargCovInv;
  NavState mNavStatePrior;
  bool mbPrior;
  const vector<MapPoint*>& GetMapPointMatches() const;
  const NavState& GetNavState() const;
  NavState& GetNavStateRef();
  const EncPreIntegrator& GetEncPreInt(void) const;
  const IMUPreintegrator& GetIMUPreInt(void) const;
  void UpdatePoseFromNS();
  void UpdateNavStatePVRFromTcw();
  cv::Mat GetCameraCenter();
  vector<MapPoint*>& GetMapPointsRef();
  vector<bool> mvbOutlier;
  cv::Mat& GetTcwRef();
  const Sophus::SE3d GetTcwCst() const;
};

class KeyFrame : public FrameBase {
 public:
  NavState mNavStatePrior;
  Matrix<double, 15, 15> mMargCovInv;
  const bool mbPrior = false;
  NavState GetNavState(void) override;
  void SetNavState(const NavState& ns) override;
  EncPreIntegrator GetEncPreInt(void) override;
  IMUPreintegrator GetIMUPreInt(void) override;
  KeyFrame* GetPrevKeyFrame(void);
  const Sophus::SE3d GetTwc() override;
  const Sophus::SE3d GetTcw() override;
  cv::Mat GetCameraCenter();
  cv::Mat GetRotation();
  cv::Mat GetTranslation();
  bool isBad() override;
  void EraseMapPointMatch(const size_t& idx) override;
  void EraseMapPointMatch(MapPoint* pMP);
  vector<MapPoint*> GetMapPointMatches() override;
  MapPoint* GetMapPoint(const size_t& idx);
  void FuseMP(size_t idx, MapPoint* pMP);
  vector<KeyFrame*> GetVectorCovisibleKeyFrames();
  unsigned long mnBALocalForKF, mnBAFixedForKF;
  void UpdateNavStatePVRFromTcw();
  NavState mNavStateGBA;
  cv::Mat mTcwGBA;
  unsigned long mnBAGlobalForKF;
};

class MapPoint {
 protected:
  typedef struct _TrackFastMatchInfo {
    float track_depth_ = INFINITY;
    bool btrack_inview_;
    static constexpr int NUM_PROJ = 3;
    std::list<float> vtrack_proj_[NUM_PROJ];
    std::list<size_t> vtrack_cami_;
    std::list<float> vtrack_viewcos_;
    std::list<int> vtrack_scalelevel_;
    unsigned long track_ref_frameid_, last_seen_frameid_;  // (FrameId = unsigned long, include/FrameBase.h:126)
    void Reset(Frame* pf = nullptr);
  } TrackFastMatchInfo;
  TrackFastMatchInfo trackinfo_;

 public:
  using Tdata = float;
  using Vector3data = Eigen::Matrix<Tdata, 3, 1>;
  MapPoint(const Vector3data& Pos, KeyFrame* pRefKF, Map* pMap);
  Vector3data GetWorldPos();
  void SetWorldPos(const Vector3data& Pos, bool block = true);
  std::map<KeyFrame*, std::set<size_t>> GetObservations();
  void EraseObservation(KeyFrame* pKF, size_t idx = -1);
  std::set<size_t> GetIndexInKeyFrame(KeyFrame* pKF);
  int Observations();
  bool IsInKeyFrame(KeyFrame* pKF, size_t idx = -1, size_t cami = -1);
  bool isBad();
  void UpdateNormalAndDepth();
  Vector3data GetNormal();
  cv::Mat GetDescriptor();
  TrackFastMatchInfo& GetTrackInfoRef();
  void IncreaseFound(int n = 1);
  void IncreaseVisible(int n = 1);
  KeyFrame* GetReferenceKeyFrame();
  Vector3data mPosGBA;
  unsigned long mnBAGlobalForKF;
  unsigned long mnId, mnBALocalForKF;
  static std::mutex mGlobalMutex;

 protected:
  float mfMinDistance, mfMaxDistance;
  Vector3data mNormalVector;
  std::mutex mMutexPos;
};

class Map {
 public:
  void InformNewChange();
  int GetLastChangeIdx();
  std::vector<KeyFrame*> GetAllKeyFrames();
  std::vector<MapPoint*> GetAllMapPoints();
  std::mutex mMutexMapUpdate;
};

void ErasePairObs(KeyFrame* pFBi, MapPoint* pMPi, size_t idx = -1);

class ORBmatcher {
 public:
  static const int TH_LOW, TH_HIGH, HISTO_LENGTH;
  enum ModeSBP { SBPFuseLater = 0x1, SBPMatchMultiCam = 0x2 };
  ORBmatcher(float nnratio = 0.6, bool checkOri = true);
  static void SearchByProjectionBase(const vector<MapPoint*>& vpMapPoints1, cv::Mat Rcrw, cv::Mat tcrw, KeyFrame* pKF,
                                     const float th_radius, const float th_bestdist, bool bCheckViewingAngle = false,
                                     const float* pbf = nullptr, int* pnfused = nullptr,
                                     char mode = (char)SBPMatchMultiCam,
                                     vector<vector<bool>>* pvbAlreadyMatched1 = nullptr,
                                     vector<set<int>>* pvnMatch1 = nullptr);
  int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3,
                         const float th_far_pts = 0);
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono,
                         const float th_far_pts = 0);
  int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                         const int ORBdist, const float th_far_pts = 0);
  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<vector<vector<size_t>>>& vMatchedPairs,
                             const bool bOnlyStereo);
  int Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th = 3.0);

 protected:
  float RadiusByViewingCos(const float& viewCos);
  float mfNNratio;
  bool mbCheckOrientation;
};

class IMUInitialization;
class Optimizer {
 public:
  template <class KeyFrame>
  int static PoseOptimization(Frame* pFrame, KeyFrame* pLastKF, const cv::Mat& gw, const bool bComputeMarg = false,
                              const bool bNoMPs = false);
  void static LocalBundleAdjustmentNavStatePRV(KeyFrame* pKF, int Nlocal, bool* pbStopFlag, Map* pMap, cv::Mat gw,
                                               bool bLarge = false, bool bRecInit = false,
                                               float th_dist_far = INFINITY);
  void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int Nlocal = 0);
  int static PoseOptimization(Frame* pFrame, Frame* pLastF = NULL);
  int static GlobalBundleAdjustmentNavStatePRV(Map* pMap, const cv::Mat& gw, int nIterations = 5, bool* pbStopFlag = NULL,
                                               const unsigned long nLoopKF = 0, const bool bRobust = true, bool bScaleOpt = false,
                                               IMUInitialization* pimu_initator = nullptr);
  void static BundleAdjustment(const std::vector<KeyFrame*>& vpKF, const std::vector<MapPoint*>& vpMP, int nIterations = 5,
                               bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true,
                               const bool bEnc = false);
  void static GlobalBundleAdjustment(Map* pMap, int nIterations = 5, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                     const bool bRobust = true, const bool bEnc = false);
};
// what INTEGRATION.md section 4 adds to include/Optimizer.h in place of the template's body
template <>
int Optimizer::PoseOptimization<Frame>(Frame*, Frame*, const cv::Mat&, const bool, const bool);
template <>
int Optimizer::PoseOptimization<KeyFrame>(Frame*, KeyFrame*, const cv::Mat&, const bool, const bool);

class System {
 public:
  enum eSensor { MONOCULAR = 0, STEREO, RGBD, NUM_SUPPORTED_CAM };
};
class LocalMapping {
 public:
  float th_far_pts_ = 0;
};
class IMUInitialization {
 public:
  cv::Mat GetGravityVec(void);
  bool GetVINSInited(void);
};

// include/Tracking.h, reduced to what shim/Tracking_hot.cc touches
class Tracking {
 public:
  void PreIntegration(const int8_t type = 0);
  bool TrackWithIMU(bool bMapUpdated);
  bool PredictNavStateByIMU(bool bMapUpdated, bool preint = true);
  bool TrackLocalMapWithIMU(bool bMapUpdated);
  enum eTrackingState { SYSTEM_NOT_READY = -1, NO_IMAGES_YET = 0, NOT_INITIALIZED = 1, OK = 2, LOST = 3, ODOMOK = 4, MAP_REUSE = 5 };
  eTrackingState mState;
  int mSensor;
  Frame mCurrentFrame;
  bool mbOnlyTracking = false;

 protected:
  void UpdateLastFrame();
  bool TrackWithMotionModel();
  void UpdateLocalMap();
  bool TrackLocalMap();
  void SearchLocalPoints();
  bool mbVO = false;
  LocalMapping* mpLocalMapper;
  IMUInitialization* mpIMUInitiator;
  vector<ORBextractor*> mpORBextractors = vector<ORBextractor*>(1, nullptr);
  ORBVocabulary* mpORBVocabulary;
  std::vector<MapPoint*> mvpLocalMapPoints;
  float mbf;
  vector<camm::Camera::Ptr> mpCameras;
  int mMaxFrames;
  float mThDepth = 10.f;
  int mnMatchesInliers;
  KeyFrame* plast_kf_ = nullptr;
  Frame mLastFrame;
  unsigned int mnLastRelocFrameId = 0;
  cv::Mat mVelocity;
  Map* mpMap;
  bool mbRelocBiasPrepare;
};

}  // namespace VIEO_SLAM
