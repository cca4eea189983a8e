// Parameters.h — singleton configuration of one simulation.
// Mirror of the part of Parameters/Parameters.h (+ CommandLineParameters.h) that the per-step loop reads:
// scalars/flags from the input (Parameters.cpp:194-459), scalar-vs-matrix medium detection by dataset shape
// (:426-459), sampling options (CommandLineParameters.cpp:264-292) and the time index (:683-702).
// The HIP-specific half (device selection, launch geometry, device constants) lives in HipParameters below,
// which replaces Parameters/CudaParameters.{h,cpp}.
#ifndef KW_HOST_PARAMETERS_H
#define KW_HOST_PARAMETERS_H
#include <cstddef>
#include <memory>
#include <string>

#include "DimensionSizes.h"
#include "InputProvider.h"
#include "kwave_hip.h"

/// Replaces class CudaParameters (Parameters/CudaParameters.h:140-146, .cpp:81-288).
class CompressHelper;

class HipParameters
{
 public:
  /// CudaParameters::selectDevice (.cpp:81-177): pick the device, create the kw_ctx.
  void selectDevice(int deviceIdx = -1);
  /// CudaParameters::setUpDeviceConstants (.cpp:238-288): fill kw_constants from Parameters and upload.
  void setUpDeviceConstants() const;
  /// launch geometry is derived inside libkwave_hip from the CU count (replaces setKernelConfiguration .cpp:195-232)
  void setKernelConfiguration() {}
  int          getDeviceIdx() const { return mDeviceIdx; }
  kw_ctx*      getContext() const { return mCtx; }
  std::string  getDeviceName() const;
  void         release();

 private:
  int     mDeviceIdx = -1;
  kw_ctx* mCtx       = nullptr;
};

class Parameters
{
 public:
  enum class SensorMaskType { kIndex = 0, kCorners = 1 };
  enum class SourceMode { kDirichlet = 0, kAdditiveNoCorrection = 1, kAdditive = 2 };
  enum class SimulationDimension { k2D, k3D };

  /// what CommandLineParameters holds for the loop (output selection + sampling start + benchmark)
  struct Options
  {
    int    deviceIdx              = -1;
    size_t samplingStartTimeIndex = 0; // -s (0-based here)
    size_t benchmarkTimeStepCount = 0; // --benchmark: overrides Nt when > 0 (Parameters.cpp:130-133)
    bool   storePressureRaw = false, storePressureRms = false, storePressureMax = false, storePressureMin = false;
    bool   storePressureMaxAll = false, storePressureMinAll = false, storePressureFinalAll = false;
    bool   storeVelocityRaw = false, storeVelocityRms = false, storeVelocityMax = false, storeVelocityMin = false;
    bool   storeVelocityMaxAll = false, storeVelocityMinAll = false, storeVelocityFinalAll = false;
    bool   storeVelocityNonStaggeredRaw = false;
    bool   storePressureC = false, storeVelocityNonStaggeredC = false, storeIntensityAvgC = false;
    bool   storeIntensityAvg = false, storeQTerm = false, storeQTermC = false; // --I_avg, --Q_term, --Q_term_c
    bool   storeVelocityC = false; // --u_c
    bool   complex40bit = false;       // --40-bit_complex
    bool   onlyPostProcessing = false; // --post: post-processing of an existing output file, no time loop
    float  frequency = 0.0f;       // --frequency [Hz], alternative to --period
    float  period = 0.0f; // --period (time steps per period)
    size_t mos = 1, harmonics = 1;
    bool   noCompressionOverlap = false;
    bool   fusedKernels = true; // MI355X fused per-step kernels (false: one launch per reference kernel)
    // Z-slab decomposition over several GPUs (one process per GPU; new with this build — the reference is single-GPU).
    // The input then describes the LOCAL slab: Nz = nzGlobal/slabRanks planes of every 3-D array, the local slice of
    // pml_z / pml_z_sgz, local (re-based) source / sensor indices; 1-D k-space operators stay global.
    size_t slabRanks = 1, slabRank = 0, nzGlobal = 0;
    // the all-to-all: by default the device library's own RCCL path — commUniqueId (KW_COMM_ID_BYTES bytes made by
    // kw_comm_unique_id on rank 0 and handed to every rank) creates the communicator; exchangeFn overrides it
    const void*    commUniqueId = nullptr;
    kw_exchange_fn exchangeFn = nullptr; // all-to-all provided by the driver (e.g. host-staged, ranks sharing a GPU)
    void*  exchangeUser = nullptr;
    kw_exchange_start_fn exchangeStartFn = nullptr; // optional split-phase pair: transposes overlap with compute
    kw_exchange_wait_fn  exchangeWaitFn  = nullptr;
    kw_exchange_piece_fn exchangePieceFn = nullptr; // optional strided form (plane chunks): pipelined slab schedule
    void*  scratch[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // optional caller-owned pipeline scratch
    bool   hasTuning = false; // schedule parameters of the device library (kw_set_tuning), else its defaults
    kw_tuning tuning{};
    bool   stepGraph = false; // steady-state step replayed from a recorded graph (small, launch-bound grids)
    // slab exchange over the device library's P2P transport: the ranks trade their export blobs through allgatherFn
    bool   commP2P = false;
    int  (*allgatherFn)(void* user, const void* mine, void* all, size_t bytes) = nullptr;
    void*  allgatherUser = nullptr;
    std::string rcclLibrary; // library the RCCL binding loads (empty: the default search)
    float  p2pEmulateLinkGbs = 0.0f, p2pEmulateLatencyUs = 0.0f; // > 0: link model instead of peers (kw_comm_p2p_emulate)
  };

  /// The parameter set the calling thread works on: the one bound by the innermost live Scope of this thread (every
  /// C-ABI entry point binds its solver handle's set), else the process-wide one (command-line program).  The reference
  /// keeps one set per process (Parameters.h:90-96); here every solver handle owns its own — parameters, device
  /// context and compression basis — so several solvers can live in one process, also on different threads.
  static Parameters& getInstance();
  static std::unique_ptr<Parameters> createDetached();
  class Scope
  {
   public:
    explicit Scope(Parameters* p);
    ~Scope();
    Scope(const Scope&)            = delete;
    Scope& operator=(const Scope&) = delete;

   private:
    Parameters* mPrevious;
  };
  ~Parameters();
  CompressHelper& getCompressHelper() { return *mCompressHelper; }

  /// Parameters::init + readScalarsFromInputFile (Parameters.cpp:113-553) on any InputProvider
  void init(const InputProvider& input, const Options& options);
  void selectDevice() { mHipParameters.selectDevice(mOptions.deviceIdx); }
  HipParameters&       getHipParameters() { return mHipParameters; }
  const HipParameters& getHipParameters() const { return mHipParameters; }
  const Options&       getOptions() const { return mOptions; }

  /// local (this rank's slab) sizes; equal to the global ones on one GPU
  DimensionSizes getFullDimensionSizes() const { return mFullDimensionSizes; }
  DimensionSizes getGlobalDimensionSizes() const { return mGlobalDimensionSizes; }
  size_t getSlabRanks() const { return mOptions.slabRanks; }
  size_t getSlabRank() const { return mOptions.slabRank; }
  bool   isSlabDecomposed() const
  {
    return mOptions.slabRanks > 1 || mOptions.exchangeFn != nullptr || mOptions.commUniqueId != nullptr || mOptions.commP2P;
  }
  DimensionSizes getReducedDimensionSizes() const { return mReducedDimensionSizes; }
  bool           isSimulation3D() const { return mGlobalDimensionSizes.is3D(); }
  bool           isSimulation2D() const { return mGlobalDimensionSizes.is2D(); }
  SimulationDimension getSimulationDimension() const
  {
    return isSimulation3D() ? SimulationDimension::k3D : SimulationDimension::k2D;
  }

  size_t getNt() const { return mNt; }
  size_t getTimeIndex() const { return mTimeIndex; }
  void   setTimeIndex(size_t t) { mTimeIndex = t; }
  void   incrementTimeIndex() { mTimeIndex++; }

  float getDt() const { return mDt; }
  float getDx() const { return mDx; }
  float getDy() const { return mDy; }
  float getDz() const { return mDz; }
  float getCRef() const { return mCRef; }
  /// PML description of the input file: not used by the loop (the pml_* vectors are), copied to the output file
  /// (Parameters.cpp:580-592); 0 when the input does not carry them
  size_t getPmlSize(int axis) const { return mPmlSize[axis]; }
  float  getPmlAlpha(int axis) const { return mPmlAlpha[axis]; }

  bool  getC0ScalarFlag() const { return mC0ScalarFlag; }
  float getC0Scalar() const { return mC0Scalar; }
  float getC2Scalar() const { return mC0Scalar * mC0Scalar; }
  bool  getRho0ScalarFlag() const { return mRho0ScalarFlag; }
  float getRho0Scalar() const { return mRho0Scalar; }
  float getRho0SgxScalar() const { return mRho0SgxScalar; }
  float getDtRho0SgxScalar() const { return mDt / mRho0SgxScalar; }
  float getDtRho0SgyScalar() const { return mDt / mRho0SgyScalar; }
  float getDtRho0SgzScalar() const { return mDt / mRho0SgzScalar; }

  size_t getNonUniformGridFlag() const { return mNonUniformGridFlag; }
  size_t getAbsorbingFlag() const { return mAbsorbingFlag; }
  size_t getNonLinearFlag() const { return mNonLinearFlag; }
  bool   getBOnAScalarFlag() const { return mBOnAScalarFlag; }
  float  getBOnAScalar() const { return mBOnAScalar; }
  bool   getAlphaCoeffScalarFlag() const { return mAlphaCoeffScalarFlag; }
  float  getAlphaCoeffScalar() const { return mAlphaCoeffScalar; }
  float  getAlphaPower() const { return mAlphaPower; }
  float  getAbsorbTauScalar() const { return mAbsorbTauScalar; }
  void   setAbsorbTauScalar(float v) { mAbsorbTauScalar = v; }
  float  getAbsorbEtaScalar() const { return mAbsorbEtaScalar; }
  void   setAbsorbEtaScalar(float v) { mAbsorbEtaScalar = v; }

  size_t     getPressureSourceFlag() const { return mPressureSourceFlag; }
  size_t     getInitialPressureSourceFlag() const { return mInitialPressureSourceFlag; }
  size_t     getTransducerSourceFlag() const { return mTransducerSourceFlag; }
  size_t     getVelocityXSourceFlag() const { return mVelocityXSourceFlag; }
  size_t     getVelocityYSourceFlag() const { return mVelocityYSourceFlag; }
  size_t     getVelocityZSourceFlag() const { return mVelocityZSourceFlag; }
  size_t     getPressureSourceIndexSize() const { return mPressureSourceIndexSize; }
  size_t     getTransducerSourceInputSize() const { return mTransducerSourceInputSize; }
  size_t     getVelocitySourceIndexSize() const { return mVelocitySourceIndexSize; }
  SourceMode getPressureSourceMode() const { return mPressureSourceMode; }
  size_t     getPressureSourceMany() const { return mPressureSourceMany; }
  SourceMode getVelocitySourceMode() const { return mVelocitySourceMode; }
  size_t     getVelocitySourceMany() const { return mVelocitySourceMany; }

  SensorMaskType getSensorMaskType() const { return mSensorMaskType; }
  size_t         getSensorMaskIndexSize() const { return mSensorMaskIndexSize; }
  size_t         getSensorMaskCornersSize() const { return mSensorMaskCornersSize; }
  size_t         getSamplingStartTimeIndex() const { return mOptions.samplingStartTimeIndex; }

  // output selection (CommandLineParameters getters of the same names)
  bool getStorePressureRawFlag() const { return mOptions.storePressureRaw; }
  bool getStorePressureRmsFlag() const { return mOptions.storePressureRms; }
  bool getStorePressureMaxFlag() const { return mOptions.storePressureMax; }
  bool getStorePressureMinFlag() const { return mOptions.storePressureMin; }
  bool getStorePressureMaxAllFlag() const { return mOptions.storePressureMaxAll; }
  bool getStorePressureMinAllFlag() const { return mOptions.storePressureMinAll; }
  bool getStorePressureFinalAllFlag() const { return mOptions.storePressureFinalAll; }
  bool getStoreVelocityRawFlag() const { return mOptions.storeVelocityRaw; }
  bool getStoreVelocityRmsFlag() const { return mOptions.storeVelocityRms; }
  bool getStoreVelocityMaxFlag() const { return mOptions.storeVelocityMax; }
  bool getStoreVelocityMinFlag() const { return mOptions.storeVelocityMin; }
  bool getStoreVelocityMaxAllFlag() const { return mOptions.storeVelocityMaxAll; }
  bool getStoreVelocityMinAllFlag() const { return mOptions.storeVelocityMinAll; }
  bool getStoreVelocityFinalAllFlag() const { return mOptions.storeVelocityFinalAll; }
  bool getStoreVelocityNonStaggeredRawFlag() const { return mOptions.storeVelocityNonStaggeredRaw; }
  bool getStorePressureCFlag() const { return mOptions.storePressureC; }
  bool getStoreVelocityNonStaggeredCFlag() const { return mOptions.storeVelocityNonStaggeredC; }
  bool getStoreIntensityAvgCFlag() const { return mOptions.storeIntensityAvgC; }
  bool getStoreIntensityAvgFlag() const { return mOptions.storeIntensityAvg; }
  bool getStoreQTermFlag() const { return mOptions.storeQTerm; }
  bool getStoreQTermCFlag() const { return mOptions.storeQTermC; }
  bool getStoreVelocityCFlag() const { return mOptions.storeVelocityC; }
  bool getOnlyPostProcessingFlag() const { return mOptions.onlyPostProcessing; }
  bool get40bitCompressionFlag() const { return mOptions.complex40bit; }
  bool getNoCompressionOverlapFlag() const { return mOptions.noCompressionOverlap; }
  float  getPeriod() const { return mOptions.period; }
  size_t getMOS() const { return mOptions.mos; }
  size_t getHarmonics() const { return mOptions.harmonics; }
  /// true when any stream needs u on the non-staggered grid (KSpaceFirstOrderSolver.cpp:1075-1078)
  bool needsShiftedVelocity() const
  {
    return mOptions.storeVelocityNonStaggeredRaw || mOptions.storeVelocityNonStaggeredC || mOptions.storeIntensityAvgC ||
           mOptions.storeIntensityAvg || mOptions.storeQTerm || mOptions.storeQTermC;
  }

 private:
  Parameters();
  Parameters(const Parameters&) = delete;

  bool                            mDetached = false; // owns its device context (released with the set)
  std::unique_ptr<CompressHelper> mCompressHelper;

  HipParameters  mHipParameters;
  Options        mOptions;
  DimensionSizes mFullDimensionSizes, mReducedDimensionSizes, mGlobalDimensionSizes;
  size_t mNt = 0, mTimeIndex = 0;
  float  mDt = 0, mDx = 0, mDy = 0, mDz = 0, mCRef = 0;
  size_t mPmlSize[3]  = {0, 0, 0};
  float  mPmlAlpha[3] = {0, 0, 0};
  bool   mC0ScalarFlag = true;
  float  mC0Scalar = 0;
  bool   mRho0ScalarFlag = true;
  float  mRho0Scalar = 0, mRho0SgxScalar = 0, mRho0SgyScalar = 0, mRho0SgzScalar = 0;
  size_t mNonUniformGridFlag = 0, mAbsorbingFlag = 0, mNonLinearFlag = 0;
  bool   mBOnAScalarFlag = true;
  float  mBOnAScalar = 0;
  bool   mAlphaCoeffScalarFlag = true;
  float  mAlphaCoeffScalar = 0, mAlphaPower = 0, mAbsorbTauScalar = 0, mAbsorbEtaScalar = 0;
  size_t mPressureSourceFlag = 0, mInitialPressureSourceFlag = 0, mTransducerSourceFlag = 0;
  size_t mVelocityXSourceFlag = 0, mVelocityYSourceFlag = 0, mVelocityZSourceFlag = 0;
  size_t mPressureSourceIndexSize = 0, mTransducerSourceInputSize = 0, mVelocitySourceIndexSize = 0;
  SourceMode mPressureSourceMode = SourceMode::kDirichlet, mVelocitySourceMode = SourceMode::kDirichlet;
  size_t mPressureSourceMany = 0, mVelocitySourceMany = 0;
  SensorMaskType mSensorMaskType = SensorMaskType::kIndex;
  size_t mSensorMaskIndexSize = 0, mSensorMaskCornersSize = 0;
};
#endif
