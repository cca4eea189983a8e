// Parameters.cpp — see Parameters.h.  Follows Parameters/Parameters.cpp:194-459 (what is read, in which order,
// how scalar-vs-matrix medium is detected) and Parameters/CudaParameters.cpp:81-177,238-288.
#include "Parameters.h"
#include <algorithm>
#include <vector>

#include <ios>
#include <stdexcept>

#include "CompressHelper.h"
#include "HipError.h"
#include "MatrixNames.h"

namespace
{
thread_local Parameters* tBound = nullptr; // the set of the solver handle whose entry point this thread is inside
}

Parameters::Parameters() : mCompressHelper(new CompressHelper()) {}

Parameters::~Parameters()
{
  if (mDetached) mHipParameters.release();
}

Parameters& Parameters::getInstance()
{
  if (tBound != nullptr) return *tBound;
  static Parameters instance;
  return instance;
}

std::unique_ptr<Parameters> Parameters::createDetached()
{
  std::unique_ptr<Parameters> p(new Parameters());
  p->mDetached = true;
  return p;
}

Parameters::Scope::Scope(Parameters* p) : mPrevious(tBound)
{
  if (p != nullptr) tBound = p;
}

Parameters::Scope::~Scope() { tBound = mPrevious; }

void Parameters::init(const InputProvider& in, const Options& options)
{
  mOptions   = options;
  mTimeIndex = 0;
  const DimensionSizes scalarSizes(1, 1, 1);

  size_t x, y, z;
  in.readScalarValue(kNxName, x);
  in.readScalarValue(kNyName, y);
  in.readScalarValue(kNzName, z);
  mFullDimensionSizes    = DimensionSizes(x, y, z);
  mReducedDimensionSizes = DimensionSizes((x / 2) + 1, y, z);
  mGlobalDimensionSizes  = mFullDimensionSizes;
  if (isSlabDecomposed())
  {
    if (mOptions.nzGlobal != z * mOptions.slabRanks || mOptions.slabRank >= mOptions.slabRanks || y % mOptions.slabRanks != 0)
      throw std::invalid_argument("Z-slab decomposition: Nz_global must equal slabRanks * local Nz and Ny must divide by slabRanks");
    mGlobalDimensionSizes.nz = mOptions.nzGlobal;
  }
  if (!isSimulation3D())
  { // 2-D (Nz == 1): what this build carries over from the 3-D path; the rest says so instead of computing nonsense
    if (z != 1 || isSlabDecomposed())
      throw std::invalid_argument("2-D simulations need Nz == 1 and a single GPU");
  }

  in.readScalarValue(kNtName, mNt);
  if (mOptions.benchmarkTimeStepCount > 0) mNt = mOptions.benchmarkTimeStepCount; // Parameters.cpp:130-133
  // Parameters.cpp:135-139: sampling must start inside the run ("-s 0" arrives here as 0 - 1, i.e. wrapped around)
  if (mOptions.samplingStartTimeIndex > mNt)
    throw std::invalid_argument("Error: The beginning of data sampling is out of the simulation time span <1, " +
                                std::to_string(mNt) + ">.");
  in.readScalarValue(kDtName, mDt);
  in.readScalarValue(kDxName, mDx);
  in.readScalarValue(kDyName, mDy);
  mDz = 0.0f;
  if (isSimulation3D()) in.readScalarValue(kDzName, mDz); // a 2-D input file has no dz (Parameters.cpp:240-243)
  in.readScalarValue(kCRefName, mCRef);
  const char* const pmlSizeNames[3]  = {"pml_x_size", "pml_y_size", "pml_z_size"};
  const char* const pmlAlphaNames[3] = {"pml_x_alpha", "pml_y_alpha", "pml_z_alpha"};
  for (int a = 0; a < 3; a++)
  {
    if (in.datasetExists(pmlSizeNames[a])) in.readScalarValue(pmlSizeNames[a], mPmlSize[a]);
    if (in.datasetExists(pmlAlphaNames[a])) in.readScalarValue(pmlAlphaNames[a], mPmlAlpha[a]);
  }

  // sensor mask (file version 1.1: Parameters.cpp:262-312)
  mSensorMaskIndexSize = mSensorMaskCornersSize = 0;
  size_t maskType = 0;
  if (in.datasetExists(kSensorMaskTypeName)) in.readScalarValue(kSensorMaskTypeName, maskType);
  mSensorMaskType = (maskType == 1) ? SensorMaskType::kCorners : SensorMaskType::kIndex;
  if (mSensorMaskType == SensorMaskType::kIndex)
  {
    if (in.datasetExists(kSensorMaskIndexName)) mSensorMaskIndexSize = in.getDatasetSize(kSensorMaskIndexName);
  }
  else
  {
    mSensorMaskCornersSize = in.getDatasetDimensionSizes(kSensorMaskCornersName).ny;
  }

  in.readScalarValue(kPressureSourceFlagName, mPressureSourceFlag);
  in.readScalarValue(kInitialPressureSourceFlagName, mInitialPressureSourceFlag);
  // 2-D input files carry neither a transducer nor a z-velocity source (Parameters.cpp:300-320)
  mTransducerSourceFlag = 0;
  mVelocityZSourceFlag  = 0;
  if (isSimulation3D() || in.datasetExists(kTransducerSourceFlagName)) in.readScalarValue(kTransducerSourceFlagName, mTransducerSourceFlag);
  in.readScalarValue(kVelocityXSourceFlagName, mVelocityXSourceFlag);
  in.readScalarValue(kVelocityYSourceFlagName, mVelocityYSourceFlag);
  if (isSimulation3D() || in.datasetExists(kVelocityZSourceFlagName)) in.readScalarValue(kVelocityZSourceFlagName, mVelocityZSourceFlag);
  if (!isSimulation3D() && (mTransducerSourceFlag != 0 || mVelocityZSourceFlag != 0))
    throw std::invalid_argument("2-D simulations have no transducer or z-velocity source");
  in.readScalarValue(kNonUniformGridFlagName, mNonUniformGridFlag);
  in.readScalarValue(kAbsorbingFlagName, mAbsorbingFlag);
  in.readScalarValue(kNonLinearFlagName, mNonLinearFlag);
  if (mNonUniformGridFlag != 0 && !isSimulation3D())
    throw std::invalid_argument("Non-uniform grids are implemented for 3-D simulations only");

  mTransducerSourceInputSize = (mTransducerSourceFlag == 0) ? 0 : in.getDatasetSize(kTransducerSourceInputName);
  mVelocitySourceIndexSize   = 0;
  if ((mTransducerSourceFlag > 0) || (mVelocityXSourceFlag > 0) || (mVelocityYSourceFlag > 0) ||
      (mVelocityZSourceFlag > 0))
    mVelocitySourceIndexSize = in.getDatasetSize(kVelocitySourceIndexName);

  auto toMode = [](size_t v, const char* what) {
    if (v > 2) throw std::ios_base::failure(std::string("Error: bad ") + what + " source mode in the input");
    return static_cast<SourceMode>(v);
  };
  if ((mVelocityXSourceFlag > 0) || (mVelocityYSourceFlag > 0) || (mVelocityZSourceFlag > 0))
  {
    in.readScalarValue(kVelocitySourceManyName, mVelocitySourceMany);
    size_t m = 0;
    in.readScalarValue(kVelocitySourceModeName, m);
    mVelocitySourceMode = toMode(m, "velocity");
  }
  else
  {
    mVelocitySourceMany = 0;
    mVelocitySourceMode = SourceMode::kDirichlet;
  }
  if (mPressureSourceFlag != 0)
  {
    in.readScalarValue(kPressureSourceManyName, mPressureSourceMany);
    size_t m = 0;
    in.readScalarValue(kPressureSourceModeName, m);
    mPressureSourceMode      = toMode(m, "pressure");
    mPressureSourceIndexSize = in.getDatasetSize(kPressureSourceIndexName);
  }
  else
  {
    mPressureSourceMode      = SourceMode::kDirichlet;
    mPressureSourceMany      = 0;
    mPressureSourceIndexSize = 0;
  }

  mAlphaCoeffScalarFlag = true;
  mAlphaCoeffScalar = mAlphaPower = mAbsorbTauScalar = mAbsorbEtaScalar = 0.0f;
  if (mAbsorbingFlag != 0)
  {
    in.readScalarValue(kAlphaPowerName, mAlphaPower);
    if (mAlphaPower == 1.0f) throw std::invalid_argument("Error: Illegal value of alpha_power (must not equal to 1.0)");
    mAlphaCoeffScalarFlag = in.getDatasetDimensionSizes(kAlphaCoeffName) == scalarSizes;
    if (mAlphaCoeffScalarFlag) in.readScalarValue(kAlphaCoeffName, mAlphaCoeffScalar);
  }
  mC0ScalarFlag = in.getDatasetDimensionSizes(kC0Name) == scalarSizes;
  if (mC0ScalarFlag) in.readScalarValue(kC0Name, mC0Scalar);
  mBOnAScalarFlag = true;
  mBOnAScalar     = 0.0f;
  if (mNonLinearFlag)
  {
    mBOnAScalarFlag = in.getDatasetDimensionSizes(kBonAName) == scalarSizes;
    if (mBOnAScalarFlag) in.readScalarValue(kBonAName, mBOnAScalar);
  }
  if (mOptions.storePressureC || mOptions.storeVelocityNonStaggeredC || mOptions.storeIntensityAvgC || mOptions.storeQTermC ||
      mOptions.storeVelocityC)
  { // Parameters.cpp:462-551: the period (in time steps) is given (--period) or found from the pressure source signal:
    // the last <= 500 samples of the middle source point (:488-512)
    if (mOptions.period > 0.0f && mOptions.frequency > 0.0f) // :468-471
      throw std::ios_base::failure("Error: --period and --frequency cannot be given together");
    if (mOptions.frequency > 0.0f) mOptions.period = 1.0f / (mOptions.frequency * mDt); // :473-477
    if (!(mOptions.period > 0.0f))
    {
      if (!in.datasetExists(kPressureSourceInputName))
        throw std::ios_base::failure("Error: compression streams need --period (> 0) or a p_source_input to derive it from");
      const DimensionSizes size = in.getDatasetDimensionSizes(kPressureSourceInputName); // (nSrc | 1, Nt_src, 1)
      std::vector<float> all(size.nElements());
      in.readFloat(kPressureSourceInputName, all.data(), all.size());
      const size_t length = std::min<size_t>(size.ny, 500);
      std::vector<float> tail(length);
      for (size_t t = 0; t < length; t++) tail[t] = all[(size.ny - length + t) * size.nx + size.nx / 2];
      mOptions.period = CompressHelper::findPeriod(tail.data(), length);
    }
    CompressHelper::getInstance().init(mOptions.period, mOptions.mos, mOptions.harmonics, true);
  }
  mRho0ScalarFlag = in.getDatasetDimensionSizes(kRho0Name) == scalarSizes;
  if (mRho0ScalarFlag)
  {
    in.readScalarValue(kRho0Name, mRho0Scalar);
    in.readScalarValue(kRho0SgxName, mRho0SgxScalar);
    in.readScalarValue(kRho0SgyName, mRho0SgyScalar);
    in.readScalarValue(kRho0SgzName, mRho0SgzScalar);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
void HipParameters::selectDevice(int deviceIdx)
{
  if (mCtx != nullptr) return; // one context per process, like the reference's single device
  kwCheck(kw_init(deviceIdx, &mCtx));
  kw_device_info info;
  kwCheck(kw_device_info_get(mCtx, &info));
  mDeviceIdx = info.device_id;
}

std::string HipParameters::getDeviceName() const
{
  if (!mCtx) return "";
  kw_device_info info;
  kwCheck(kw_device_info_get(mCtx, &info));
  return std::string(info.name) + " (" + info.arch + ")";
}

void HipParameters::release()
{
  if (mCtx) kw_destroy(mCtx);
  mCtx = nullptr;
}

void HipParameters::setUpDeviceConstants() const
{
  const Parameters& params = Parameters::getInstance();
  const DimensionSizes full = params.getFullDimensionSizes(), red = params.getReducedDimensionSizes();
  kw_constants k{};
  k.nx = static_cast<uint32_t>(full.nx);
  k.ny = static_cast<uint32_t>(full.ny);
  k.nz = static_cast<uint32_t>(full.nz);
  k.n_elements = static_cast<uint32_t>(full.nElements());
  k.nx_complex = static_cast<uint32_t>(red.nx);
  k.ny_complex = static_cast<uint32_t>(red.ny);
  k.nz_complex = static_cast<uint32_t>(red.nz);
  k.n_elements_complex = static_cast<uint32_t>(red.nElements());
  const DimensionSizes global = params.getGlobalDimensionSizes();
  k.fft_divider   = 1.0f / global.nElements(); // 1/N of the whole grid also on a slab
  k.fft_divider_x = 1.0f / global.nx;
  k.fft_divider_y = 1.0f / global.ny;
  k.fft_divider_z = 1.0f / global.nz;
  k.dt      = params.getDt();
  k.dt_by_2 = params.getDt() * 2.0f;
  k.c2      = params.getC2Scalar();
  k.rho0    = params.getRho0Scalar();
  k.dt_rho0 = params.getRho0Scalar() * params.getDt();
  if (params.getRho0ScalarFlag())
  {
    k.dt_rho0_sgx = params.getDtRho0SgxScalar();
    k.dt_rho0_sgy = params.getDtRho0SgyScalar();
    k.dt_rho0_sgz = params.getDtRho0SgzScalar();
  }
  k.b_on_a     = params.getBOnAScalar();
  k.absorb_tau = params.getAbsorbTauScalar();
  k.absorb_eta = params.getAbsorbEtaScalar();
  k.pressure_source_size = static_cast<uint32_t>(params.getPressureSourceIndexSize());
  k.pressure_source_mode = static_cast<uint32_t>(params.getPressureSourceMode());
  k.pressure_source_many = static_cast<uint32_t>(params.getPressureSourceMany());
  k.velocity_source_size = static_cast<uint32_t>(params.getVelocitySourceIndexSize());
  k.velocity_source_mode = static_cast<uint32_t>(params.getVelocitySourceMode());
  k.velocity_source_many = static_cast<uint32_t>(params.getVelocitySourceMany());
  kwCheck(kw_set_constants(mCtx, &k));
}
