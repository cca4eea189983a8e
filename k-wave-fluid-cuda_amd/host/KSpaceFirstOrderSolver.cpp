// KSpaceFirstOrderSolver.cpp — see KSpaceFirstOrderSolver.h.  Sequencing follows
// KSpaceSolver/KSpaceFirstOrderSolver.cpp:864-943 (loop), :2087-2396 (step pieces), :2404-2703 (generators).
#include "KSpaceFirstOrderSolver.h"
#include <cstdlib>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <limits>
#include <string>

#include "HipError.h"
#include "SolverHipKernels.h"

using SD = Parameters::SimulationDimension;
using MI = MatrixContainer::MatrixIdx;

namespace
{
struct PhaseTimer
{ // adds the lifetime of the object to one of the solver's phase counters
  explicit PhaseTimer(double& acc) : mAcc(acc), mStart(std::chrono::steady_clock::now()) {}
  ~PhaseTimer() { mAcc += std::chrono::duration<double>(std::chrono::steady_clock::now() - mStart).count(); }
  double& mAcc;
  std::chrono::steady_clock::time_point mStart;
};
} // namespace

KSpaceFirstOrderSolver::KSpaceFirstOrderSolver() : mParameters(Parameters::getInstance()) {}
KSpaceFirstOrderSolver::~KSpaceFirstOrderSolver() { freeMemory(); }

void KSpaceFirstOrderSolver::allocateMemory()
{
  mMatrixContainer.init();
  mMatrixContainer.createMatrices();
  mOutputStreamContainer.init(mMatrixContainer);
}

void KSpaceFirstOrderSolver::freeMemory()
{
  releaseFusedPipeline();
  mOutputStreamContainer.freeStreams();
  mMatrixContainer.freeMatrices();
  HipFftComplexMatrix::destroyAllPlansAndStaticData();
  mPrepared = false;
}

void KSpaceFirstOrderSolver::loadInputData(const InputProvider& input)
{
  PhaseTimer timer(mPhaseTime[0]);
  mMatrixContainer.loadDataFromInputFile(input);
  mOutputStreamContainer.createStreams();
}

void KSpaceFirstOrderSolver::initializeFftPlans()
{ // KSpaceFirstOrderSolver.cpp:747-777
  const DimensionSizes dims = mParameters.getFullDimensionSizes();
  kw_ctx* ctx = mParameters.getHipParameters().getContext();
  int fusedOk = 0;
  const Parameters::Options& opt = mParameters.getOptions();
  if (opt.hasTuning) kwCheck(kw_set_tuning(ctx, &opt.tuning));
  const bool libraryExchange = mParameters.isSlabDecomposed() && (opt.commUniqueId != nullptr || opt.commP2P);
  if (libraryExchange && (opt.exchangeFn != nullptr || opt.exchangeStartFn != nullptr))
    throw std::invalid_argument("Z-slab decomposition: give either the device library's exchange (commUniqueId / commP2P) or exchange callbacks");
  if (mParameters.isSlabDecomposed() && opt.commUniqueId != nullptr)
  { // the device library's own all-to-all: RCCL communicator + communication stream on the context (collective call)
    kwCheck(kw_comm_init_with(ctx, opt.rcclLibrary.empty() ? nullptr : opt.rcclLibrary.c_str(), static_cast<uint32_t>(opt.slabRanks),
                              static_cast<uint32_t>(opt.slabRank), opt.commUniqueId));
    mOwnsComm = true;
  }
  if (mParameters.isSlabDecomposed() && opt.commP2P)
  { // device-initiated transport: mapped peer buffers, one store kernel per exchange (connected below, once the buffers exist)
    if (opt.allgatherFn == nullptr && opt.slabRanks > 1 && !(opt.p2pEmulateLinkGbs > 0.0f))
      throw std::invalid_argument("Z-slab decomposition over the P2P transport needs allgatherFn (the ranks trade their buffer handles through it)");
    kwCheck(kw_comm_init_p2p(ctx, static_cast<uint32_t>(opt.slabRanks), static_cast<uint32_t>(opt.slabRank)));
    mOwnsComm = true;
  }
  if (mParameters.isSlabDecomposed())
    kwCheck(kw_fused_set_slab(ctx, static_cast<uint32_t>(opt.slabRanks), static_cast<uint32_t>(opt.slabRank),
                              static_cast<uint32_t>(opt.nzGlobal), opt.exchangeFn, opt.exchangeUser));
  if (mParameters.isSlabDecomposed() && opt.exchangeStartFn != nullptr)
    kwCheck(kw_fused_set_slab_async(ctx, opt.exchangeStartFn, opt.exchangeWaitFn));
  if (mParameters.isSlabDecomposed() && opt.exchangePieceFn != nullptr)
    kwCheck(kw_fused_set_slab_pieces(ctx, opt.exchangePieceFn, opt.exchangeStartFn != nullptr ? nullptr : opt.exchangeWaitFn));
  if (opt.fusedKernels || mParameters.isSlabDecomposed()) kwCheck(kw_fused_supported(ctx, &fusedOk));
  mFused = (fusedOk != 0);
  if (mParameters.isSlabDecomposed() && !mFused)
    throw std::invalid_argument("Z-slab decomposition needs the fused pipeline (supported line lengths; Ny and Nz divisible by the rank count)");
  if (mFused)
  { // hand-written FFT passes: no library plans needed for the 3-D transforms
    if (opt.scratch[0] != nullptr) kwCheck(kw_fused_create_with_scratch(ctx, opt.scratch, opt.scratch + 3));
    else kwCheck(kw_fused_create(ctx));
    if (mParameters.isSlabDecomposed() && opt.commP2P && opt.p2pEmulateLinkGbs > 0.0f)
      kwCheck(kw_comm_p2p_emulate(ctx, opt.p2pEmulateLinkGbs, opt.p2pEmulateLatencyUs));
    else if (mParameters.isSlabDecomposed() && opt.commP2P)
    { // every rank publishes where its exchange buffers are; all ranks map all of them
      std::vector<char> mine(KW_COMM_P2P_BLOB_BYTES), all(KW_COMM_P2P_BLOB_BYTES * opt.slabRanks);
      kwCheck(kw_comm_p2p_export(ctx, mine.data(), mine.size()));
      if (opt.slabRanks == 1) all = mine;
      else if (opt.allgatherFn(opt.allgatherUser, mine.data(), all.data(), mine.size()) != 0)
        throw std::runtime_error("Z-slab decomposition: the caller's allgather of the P2P buffer handles failed");
      kwCheck(kw_comm_p2p_connect(ctx, all.data()));
    }
  }
  else
  {
    HipFftComplexMatrix::createR2CFftPlanND(dims);
    HipFftComplexMatrix::createC2RFftPlanND(dims);
  }
  if (mParameters.needsShiftedVelocity() && !mParameters.isSlabDecomposed())
  { // (slab runs shift through the fused pipeline only: the z lines cross the slabs, kw_fused_shift_velocity)
    HipFftComplexMatrix::createR2CFftPlan1DX(dims);
    HipFftComplexMatrix::createR2CFftPlan1DY(dims);
    if (mParameters.isSimulation3D()) HipFftComplexMatrix::createR2CFftPlan1DZ(dims);
  }
}

void KSpaceFirstOrderSolver::prepare()
{
  if (mPrepared) return;
  PhaseTimer timer(mPhaseTime[1]);
  mParameters.getHipParameters().setUpDeviceConstants(); // dims first: plans need them
  initializeFftPlans();
  preProcessing<SD::k3D>();
  mParameters.getHipParameters().setKernelConfiguration();
  mParameters.getHipParameters().setUpDeviceConstants(); // again: tau/eta scalars exist only after preProcessing
  mMatrixContainer.copyMatricesToDevice();               // KSpaceFirstOrderSolver.cpp:880
  if (mFused) initializeFusedPipeline();
  // host twins of the big arrays are not needed during the loop
  for (auto& rec : mMatrixContainer.records())
    if (rec.second.matrixType != MatrixRecord::MatrixType::kIndex)
      static_cast<BaseFloatMatrix*>(rec.second.matrixPtr)->freeHostData();
  mPrepared = true;
}

void KSpaceFirstOrderSolver::compute()
{
  prepare();
  computeMainLoop<SD::k3D>();
  postProcessing<SD::k3D>();
}

void KSpaceFirstOrderSolver::finish()
{
  if (mParameters.getTimeIndex() > mParameters.getSamplingStartTimeIndex()) mOutputStreamContainer.flushRawStreams();
  postProcessing<SD::k3D>();
}

// ---------------------------------------------------------------------------------------------------------------------
template<SD sd> void KSpaceFirstOrderSolver::preProcessing()
{ // KSpaceFirstOrderSolver.cpp:784-857
  if (mMatrixContainer.has(MI::kSensorMaskIndex)) index(MI::kSensorMaskIndex).recomputeIndicesToCPP();
  if (mMatrixContainer.has(MI::kSensorMaskCorners)) index(MI::kSensorMaskCorners).recomputeIndicesToCPP();
  if ((mParameters.getTransducerSourceFlag() != 0) || (mParameters.getVelocityXSourceFlag() != 0) ||
      (mParameters.getVelocityYSourceFlag() != 0) || (mParameters.getVelocityZSourceFlag() != 0))
    index(MI::kVelocitySourceIndex).recomputeIndicesToCPP();
  if (mParameters.getTransducerSourceFlag() != 0) index(MI::kDelayMask).recomputeIndicesToCPP();
  if (mParameters.getPressureSourceFlag() != 0) index(MI::kPressureSourceIndex).recomputeIndicesToCPP();
  // an index outside the grid would be an out-of-bounds access in a gather / scatter kernel: refuse it here (the
  // reference trusts the file)
  const DimensionSizes grid = mParameters.getFullDimensionSizes();
  auto checkIndices = [&](MI idx, const char* name) {
    if (!mMatrixContainer.has(idx)) return;
    const IndexMatrix& m = index(idx);
    const size_t* v = m.getHostData();
    for (size_t i = 0; i < m.size(); i++)
      if (v[i] >= grid.nElements())
        throw std::invalid_argument(std::string(name) + ": index " + std::to_string(v[i] + 1) + " (1-based) lies outside the " +
                                    std::to_string(grid.nx) + " x " + std::to_string(grid.ny) + " x " + std::to_string(grid.nz) + " grid");
  };
  checkIndices(MI::kSensorMaskIndex, "sensor_mask_index");
  if ((mParameters.getTransducerSourceFlag() != 0) || (mParameters.getVelocityXSourceFlag() != 0) ||
      (mParameters.getVelocityYSourceFlag() != 0) || (mParameters.getVelocityZSourceFlag() != 0))
    checkIndices(MI::kVelocitySourceIndex, "u_source_index");
  if (mParameters.getPressureSourceFlag() != 0) checkIndices(MI::kPressureSourceIndex, "p_source_index");
  if (mParameters.getTransducerSourceFlag() != 0)
  { // every point reads signal[delay + step] for the steps the source is on (SolverCudaKernels.cu:430-449)
    const IndexMatrix& d = index(MI::kDelayMask);
    const size_t steps  = std::min(mParameters.getNt(), mParameters.getTransducerSourceFlag());
    const size_t length = real(MI::kTransducerSourceInput).size();
    for (size_t i = 0; i < d.size(); i++)
      if (d.getHostData()[i] + steps > length)
        throw std::invalid_argument("delay_mask: delay " + std::to_string(d.getHostData()[i] + 1) + " plus " + std::to_string(steps) +
                                    " steps runs past the end of transducer_source_input (" + std::to_string(length) + " samples)");
  }
  if (mMatrixContainer.has(MI::kSensorMaskCorners))
  {
    const IndexMatrix& m = index(MI::kSensorMaskCorners);
    for (size_t c = 0; c < m.getDimensionSizes().ny; c++)
    {
      const DimensionSizes a = m.getTopLeftCorner(c), b = m.getBottomRightCorner(c);
      if (a.nx > b.nx || a.ny > b.ny || a.nz > b.nz || b.nx >= grid.nx || b.ny >= grid.ny || b.nz >= grid.nz)
        throw std::invalid_argument("sensor_mask_corners: cuboid " + std::to_string(c + 1) + " is empty or lies outside the grid");
    }
  }

  if (mParameters.getNonUniformGridFlag() != 0) generateInitialDenisty();
  else if (!mParameters.getRho0ScalarFlag())
  {
    real(MI::kDtRho0Sgx).scalarDividedBy(mParameters.getDt());
    real(MI::kDtRho0Sgy).scalarDividedBy(mParameters.getDt());
    real(MI::kDtRho0Sgz).scalarDividedBy(mParameters.getDt());
  }
  if (mParameters.getAbsorbingFlag() != 0)
  {
    generateKappaAndNablas();
    generateTauAndEta();
  }
  else
  {
    generateKappa();
  }
  if (mMatrixContainer.has(MI::kSourceKappa)) generateSourceKappa();
  computeC2();
}

template<SD sd> void KSpaceFirstOrderSolver::computeMainLoop()
{ // KSpaceFirstOrderSolver.cpp:864-943
  runTimeSteps(mParameters.getNt() - mParameters.getTimeIndex());
  if (mParameters.getTimeIndex() > mParameters.getSamplingStartTimeIndex()) mOutputStreamContainer.flushRawStreams();
}

void KSpaceFirstOrderSolver::runTimeSteps(size_t nSteps)
{
  prepare();
  PhaseTimer timer(mPhaseTime[2]);
  mPressureInScratch = false;
  for (size_t s = 0; s < nSteps && mParameters.getTimeIndex() < mParameters.getNt(); s++)
  {
    const size_t timeIndex = mParameters.getTimeIndex();
    // fused pipeline: the velocity stage may hand the x-spectra of u straight to the density stage when no velocity /
    // transducer source writes u in between this step
    mVelocityChained = mFused && !(mParameters.getVelocityXSourceFlag() > timeIndex) &&
                       !(mParameters.getVelocityYSourceFlag() > timeIndex) &&
                       !(mParameters.getVelocityZSourceFlag() > timeIndex) &&
                       !(mParameters.getTransducerSourceFlag() > timeIndex);
    // steady state of the fused pipeline: no source writes a field this step and the step starts from the chained
    // spectrum of p — then the step is the same sequence of launches with the same arguments every time, and its host
    // and pipeline state on exit equals that on entry: it can be replayed from a recorded graph
    const bool steady = mUseStepGraph && !mStepGraphFailed && mVelocityChained && mPressureInScratch && (timeIndex > 0) &&
                        !(mParameters.getPressureSourceFlag() > timeIndex) &&
                        !kw_profile_enabled(mParameters.getHipParameters().getContext());
    auto stages = [&]() {
      computeVelocity<SD::k3D>();
      computeVelocityGradient<SD::k3D>();
      if (mParameters.getNonLinearFlag()) computeDensityNonliner<SD::k3D>();
      else computeDensityLinear<SD::k3D>();
      if (mParameters.getNonLinearFlag()) computePressureNonlinear<SD::k3D>();
      else computePressureLinear<SD::k3D>();
    };
    if (steady)
    {
      kw_ctx* ctx = mParameters.getHipParameters().getContext();
      if (mStepGraph == nullptr)
      {
        if (kw_graph_begin(ctx) == KW_OK)
        {
          try { stages(); }
          catch (...) { kw_graph* junk = nullptr; (void)kw_graph_end(ctx, &junk); (void)kw_graph_destroy(ctx, junk); throw; }
          kwCheck(kw_graph_end(ctx, &mStepGraph));
        }
        else mStepGraphFailed = true; // e.g. per-call profiling is on: run eagerly
      }
      if (mStepGraph != nullptr) kwCheck(kw_graph_launch(ctx, mStepGraph));
      else stages();
      storeSensorData();
      mParameters.incrementTimeIndex();
      continue;
    }
    computeVelocity<SD::k3D>();
    addVelocitySource();
    if (mParameters.getTransducerSourceFlag() > timeIndex) SolverHipKernels::addTransducerSource(mMatrixContainer);
    computeVelocityGradient<SD::k3D>();
    if (mParameters.getNonLinearFlag()) computeDensityNonliner<SD::k3D>();
    else computeDensityLinear<SD::k3D>();
    addPressureSource<SD::k3D>();
    if (mParameters.getNonLinearFlag()) computePressureNonlinear<SD::k3D>();
    else computePressureLinear<SD::k3D>();
    if ((timeIndex == 0) && (mParameters.getInitialPressureSourceFlag() == 1)) addInitialPressureSource<SD::k3D>();
    storeSensorData();
    mParameters.incrementTimeIndex();
  }
}

template<SD sd> void KSpaceFirstOrderSolver::postProcessing()
{ // KSpaceFirstOrderSolver.cpp:950-1053 (streams part); p_final/u_final stay on the device until asked for
  using OI = OutputStreamContainer::OutputStreamIdx;
  {
    PhaseTimer loop(mPhaseTime[2]); // launches still in flight belong to the time loop
    kwCheck(kw_sync(mParameters.getHipParameters().getContext()));
  }
  PhaseTimer timer(mPhaseTime[3]);
  // average intensity from the stored p and u_non_staggered series (:982-987)
  if (mParameters.getStoreQTermFlag() || mParameters.getStoreIntensityAvgFlag()) computeAverageIntensities();
  // --post: the intensity of the compression path from the stored coefficient frames (:989-996); a simulating run has
  // accumulated it frame by frame already
  if (mParameters.getOnlyPostProcessingFlag() && (mParameters.getStoreQTermCFlag() || mParameters.getStoreIntensityAvgCFlag()))
    computeAverageIntensitiesC();
  mOutputStreamContainer.postProcessStreams();
  // volume rate of heat deposition from the average intensity (:1001-1021)
  if (mParameters.getStoreQTermFlag())
    computeQTerm(OI::kIntensityXAvg, OI::kIntensityYAvg, OI::kIntensityZAvg, OI::kQTerm);
  if (mParameters.getStoreQTermCFlag())
    computeQTerm(OI::kIntensityXAvgC, OI::kIntensityYAvgC, OI::kIntensityZAvgC, OI::kQTermC);
  mOutputStreamContainer.closeStreams();
}

// ---------------------------------------------------------------------------------------------------------------------
// Post-processing.  Same arithmetic as the reference; the spectra stay on the device (the reference copies every
// spectrum to the host for the multiply and back, :1437-1487, :1941-2004).
// ---------------------------------------------------------------------------------------------------------------------
namespace
{
struct DeviceBuffer
{ // scoped device allocation
  DeviceBuffer(kw_ctx* c, size_t bytes) : ctx(c) { kwCheck(kw_malloc(ctx, bytes, &ptr)); }
  ~DeviceBuffer() { if (ptr) kw_free(ctx, ptr); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  float* f() const { return static_cast<float*>(ptr); }
  kw_ctx* ctx;
  void*   ptr = nullptr;
};
} // namespace

std::vector<size_t> KSpaceFirstOrderSolver::sensorGridIndices()
{
  std::vector<size_t> idx;
  if (mMatrixContainer.has(MI::kSensorMaskIndex))
  {
    const IndexMatrix& m = index(MI::kSensorMaskIndex);
    idx.assign(m.getHostData(), m.getHostData() + m.size());
    return idx;
  }
  const IndexMatrix&   m    = index(MI::kSensorMaskCorners);
  const DimensionSizes dims = mParameters.getFullDimensionSizes();
  for (size_t c = 0; c < m.getDimensionSizes().ny; c++)
  { // buffer order of a cuboid: x fastest, then y, then z (:1822-1840)
    const DimensionSizes a = m.getTopLeftCorner(c), b = m.getBottomRightCorner(c);
    for (size_t z = a.nz; z <= b.nz; z++)
      for (size_t y = a.ny; y <= b.ny; y++)
        for (size_t x = a.nx; x <= b.nx; x++) idx.push_back((z * dims.ny + y) * dims.nx + x);
  }
  return idx;
}

void KSpaceFirstOrderSolver::computeAverageIntensities()
{ // :1231-1534: u is sampled half a step after p; its series is advanced by half a step through its spectrum along
  // time, then I = mean over the stored steps of p * u
  using OI = OutputStreamContainer::OutputStreamIdx;
  kw_ctx* ctx = mParameters.getHipParameters().getContext();
  BaseOutputStream* ps = mOutputStreamContainer.get(OI::kPressureRaw);
  const OI ui[3] = {OI::kVelocityXNonStaggeredRaw, OI::kVelocityYNonStaggeredRaw, OI::kVelocityZNonStaggeredRaw};
  const OI ii[3] = {OI::kIntensityXAvg, OI::kIntensityYAvg, OI::kIntensityZAvg};
  if (ps == nullptr) throw std::runtime_error("computeAverageIntensities: the raw pressure stream is missing");
  const size_t steps = ps->sampledSteps(), points = ps->size();
  if (steps < 2 || points == 0) return; // nothing stored (sampling never started): the intensities stay zero
  ps->loadSeries(); // series streamed to the output file come back from there (the reference re-reads its datasets too, :1331-1370)

  // phase factors of the half-step shift (:1253-1260)
  using FloatComplex = std::complex<float>;
  const float  pi2 = static_cast<float>(M_PI) * 2.0f;
  const size_t stepsComplex = steps / 2 + 1;
  std::vector<FloatComplex> kx(stepsComplex);
  for (size_t i = 0; i < stepsComplex; i++)
  {
    const ssize_t shift = ssize_t((i + (steps / 2)) % steps - (steps / 2));
    kx[i] = std::exp(FloatComplex(0.0f, 1.0f) * (pi2 * 0.5f) * (float(shift) / float(steps)));
  }
  DeviceBuffer dShift(ctx, stepsComplex * sizeof(FloatComplex));
  kwCheck(kw_memcpy_h2d(ctx, dShift.ptr, kx.data(), stepsComplex * sizeof(FloatComplex)));

  // blocks of sensor points: p, u and the spectrum of u (~ 3.1 arrays of block x steps floats) stay below ~1 GB
  const size_t budget = size_t(80) << 20; // floats per array
  size_t block = std::max<size_t>(1, std::min(points, budget / steps));
  DeviceBuffer dP(ctx, block * steps * sizeof(float)), dU(ctx, block * steps * sizeof(float)), dI(ctx, block * sizeof(float));
  std::vector<float> pack(block * steps);
  auto upload = [&](const std::vector<float>& series, size_t first, size_t n, float* dst) {
    for (size_t s = 0; s < steps; s++) std::copy_n(series.data() + s * points + first, n, pack.data() + s * n);
    kwCheck(kw_memcpy_h2d(ctx, dst, pack.data(), n * steps * sizeof(float)));
  };
  for (size_t first = 0; first < points; first += block)
  {
    const size_t n = std::min(block, points - first);
    upload(ps->dataset(), first, n, dP.f());
    for (int a = 0; a < (mParameters.isSimulation3D() ? 3 : 2); a++)
    {
      BaseOutputStream* us = mOutputStreamContainer.get(ui[a]);
      auto* is = dynamic_cast<PostProcessedOutputStream*>(mOutputStreamContainer.get(ii[a]));
      if (us == nullptr || is == nullptr) throw std::runtime_error("computeAverageIntensities: stream missing");
      if (first == 0) us->loadSeries();
      upload(us->dataset(), first, n, dU.f());
      kwCheck(kw_time_shift_series(ctx, dU.f(), dShift.f(), steps, n));
      kwCheck(kw_intensity_avg(ctx, dI.f(), dP.f(), dU.f(), steps, n));
      kwCheck(kw_memcpy_d2h(ctx, is->data().data() + first, dI.f(), n * sizeof(float)));
    }
  }
  ps->releaseSeries();
  for (int a = 0; a < 3; a++)
    if (BaseOutputStream* us = mOutputStreamContainer.get(ui[a])) us->releaseSeries();
}

void KSpaceFirstOrderSolver::computeAverageIntensitiesC()
{ // :1541-1780: frame by frame, I += sum_h Re(P conj(U)) / 2; the mean over the frames is taken by the stream's postProcess
  using OI = OutputStreamContainer::OutputStreamIdx;
  BaseOutputStream* pc = mOutputStreamContainer.get(OI::kPressureC);
  const OI ui[3] = {OI::kVelocityXNonStaggeredC, OI::kVelocityYNonStaggeredC, OI::kVelocityZNonStaggeredC};
  const OI ii[3] = {OI::kIntensityXAvgC, OI::kIntensityYAvgC, OI::kIntensityZAvgC};
  if (pc == nullptr) throw std::runtime_error("computeAverageIntensitiesC: the stream p_c is missing");
  pc->loadSeries();
  const size_t frames = pc->sampledSteps(), floats = pc->size();
  for (int a = 0; a < (mParameters.isSimulation3D() ? 3 : 2); a++)
  {
    BaseOutputStream* uc = mOutputStreamContainer.get(ui[a]);
    auto* is = dynamic_cast<IntensityAvgCOutputStream*>(mOutputStreamContainer.get(ii[a]));
    if (uc == nullptr || is == nullptr) throw std::runtime_error("computeAverageIntensitiesC: stream missing");
    uc->loadSeries();
    if (uc->sampledSteps() != frames || uc->size() != floats) throw std::runtime_error("computeAverageIntensitiesC: p_c and u_c series differ in shape");
    for (size_t f = 0; f < frames; f++) is->accumulateStoredFrames(pc->dataset().data() + f * floats, uc->dataset().data() + f * floats);
    uc->releaseSeries();
  }
  pc->releaseSeries();
}

void KSpaceFirstOrderSolver::postProcessStoredOutput()
{ // compute() of a --post run (:373-415): no time loop, straight to the post-processing of what the output file holds
  prepare();
  mParameters.setTimeIndex(mParameters.getNt());
  postProcessing<SD::k3D>();
}

void KSpaceFirstOrderSolver::computeQTerm(OutputStreamContainer::OutputStreamIdx intensityX,
                                          OutputStreamContainer::OutputStreamIdx intensityY,
                                          OutputStreamContainer::OutputStreamIdx intensityZ,
                                          OutputStreamContainer::OutputStreamIdx qTerm)
{ // :1783-2080: Q = -div(I_avg), each derivative taken spectrally along its own axis of the full grid (zero outside
  // the sensor mask)
  using FloatComplex = std::complex<float>;
  kw_ctx* ctx = mParameters.getHipParameters().getContext();
  const DimensionSizes dims = mParameters.getFullDimensionSizes();
  const std::vector<size_t> where = sensorGridIndices();
  auto* q = dynamic_cast<PostProcessedOutputStream*>(mOutputStreamContainer.get(qTerm));
  if (q == nullptr) throw std::runtime_error("computeQTerm: the Q-term stream is missing");
  const OutputStreamContainer::OutputStreamIdx ii[3] = {intensityX, intensityY, intensityZ};
  RealMatrix* grid[3] = {&getTemp1RealND(), &getTemp2RealND(), &getTemp3RealND()};
  HipFftComplexMatrix& tempShift = fft(MI::kTempHipFftShift);
  const float  pi2 = static_cast<float>(M_PI) * 2.0f;
  const DimensionSizes global = mParameters.getGlobalDimensionSizes(); // z lines have the global length in slab mode
  const size_t n[3] = {dims.nx, dims.ny, global.nz};
  const float  d[3] = {mParameters.getDx(), mParameters.getDy(), mParameters.getDz()};
  const int    axes = mParameters.isSimulation3D() ? 3 : 2;
  for (int a = 0; a < axes; a++)
  {
    BaseOutputStream* is = mOutputStreamContainer.get(ii[a]);
    if (is == nullptr || is->dataset().size() != where.size()) throw std::runtime_error("computeQTerm: intensity stream missing");
    float* host = grid[a]->getHostData();
    std::fill_n(host, dims.nElements(), 0.0f);
    for (size_t i = 0; i < where.size(); i++) host[where[i]] = is->dataset()[i];
    grid[a]->copyToDevice();
    // i*k of this axis (:1905-1921); the 1/N of the transform pair is applied with it (:1945-1958)
    const size_t nc = n[a] / 2 + 1;
    std::vector<FloatComplex> k(nc);
    for (size_t i = 0; i < nc; i++)
    {
      const ssize_t shift = ssize_t((i + (n[a] / 2)) % n[a] - (n[a] / 2));
      k[i] = FloatComplex(0.0f, 1.0f) * (pi2 / d[a]) * (float(shift) / float(n[a]));
    }
    if (mFused)
    { // one kernel per axis on the hand-written passes (and through the exchange for z lines that cross slabs): the
      // half-length i*k extended to the full-length Hermitian filter R2C -> multiply -> C2R amounts to, 1/N folded in
      std::vector<FloatComplex> full(n[a], FloatComplex(0.0f, 0.0f));
      const float divider = 1.0f / static_cast<float>(n[a]);
      for (size_t i = 1; i < (n[a] + 1) / 2; i++)
      {
        full[i]        = k[i] * divider;
        full[n[a] - i] = std::conj(full[i]);
      } // (bins 0 and N/2: i*k is imaginary there, C2R keeps the real part: zero)
      DeviceBuffer df(ctx, n[a] * sizeof(FloatComplex));
      kwCheck(kw_memcpy_h2d(ctx, df.ptr, full.data(), n[a] * sizeof(FloatComplex)));
      kwCheck(kw_fused_shift_velocity(ctx, a, grid[a]->getDeviceData(), grid[a]->getDeviceData(), df.f()));
      kwCheck(kw_sync(ctx)); // df goes out of scope
      continue;
    }
    DeviceBuffer dk(ctx, nc * sizeof(FloatComplex));
    kwCheck(kw_memcpy_h2d(ctx, dk.ptr, k.data(), nc * sizeof(FloatComplex)));
    kwCheck(kw_fft_r2c_1d(ctx, a, grid[a]->getDeviceData(), tempShift.getDeviceData()));
    kwCheck(kw_compute_velocity_shift(ctx, a, tempShift.getDeviceData(), dk.f()));
    kwCheck(kw_fft_c2r_1d(ctx, a, tempShift.getDeviceData(), grid[a]->getDeviceData()));
  }
  kwCheck(kw_q_term_sum(ctx, grid[0]->getDeviceData(), grid[0]->getDeviceData(), grid[1]->getDeviceData(),
                        (axes == 3) ? grid[2]->getDeviceData() : nullptr, dims.nElements()));
  grid[0]->copyFromDevice();
  const float* host = grid[0]->getHostData();
  for (size_t i = 0; i < where.size(); i++) q->data()[i] = host[where[i]];
}

void KSpaceFirstOrderSolver::storeSensorData()
{ // KSpaceFirstOrderSolver.cpp:1060-1093
  if (mOutputStreamContainer.empty()) return;
  if (mParameters.getTimeIndex() >= mParameters.getSamplingStartTimeIndex())
  {
    if (mParameters.getTimeIndex() > mParameters.getSamplingStartTimeIndex()) mOutputStreamContainer.flushRawStreams();
    if (mParameters.needsShiftedVelocity()) computeShiftedVelocity<SD::k3D>();
    mOutputStreamContainer.sampleStreams();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
template<SD sd> void KSpaceFirstOrderSolver::computeVelocity()
{ // :2087-2119
  if (mFused)
  {
    const MatrixContainer& c = mMatrixContainer;
    kwCheck(kw_fused_velocity(mParameters.getHipParameters().getContext(), getP().getDeviceData(),
                              real(MI::kUxSgx).getDeviceData(), real(MI::kUySgy).getDeviceData(),
                              real(MI::kUzSgz).getDeviceData(), c.realDeviceOrNull(MI::kDtRho0Sgx),
                              c.realDeviceOrNull(MI::kDtRho0Sgy), c.realDeviceOrNull(MI::kDtRho0Sgz),
                              real(MI::kPmlXSgx).getDeviceData(), real(MI::kPmlYSgy).getDeviceData(),
                              real(MI::kPmlZSgz).getDeviceData(), mKappaPadded,
                              c.getMatrix<ComplexMatrix>(MI::kDdxKShiftPosR).getDeviceData(),
                              c.getMatrix<ComplexMatrix>(MI::kDdyKShiftPos).getDeviceData(),
                              c.getMatrix<ComplexMatrix>(MI::kDdzKShiftPos).getDeviceData(),
                              (mVelocityChained ? KW_FUSED_CHAIN_U : 0) | (mPressureInScratch ? KW_FUSED_P_IN_SCRATCH : 0)));
    mPressureInScratch = false;
    return;
  }
  getTempHipFftX().computeR2CFftND(getP());
  SolverHipKernels::computePressureGradient<sd>(mMatrixContainer);
  getTempHipFftX().computeC2RFftND(getTemp1RealND());
  getTempHipFftY().computeC2RFftND(getTemp2RealND());
  getTempHipFftZ().computeC2RFftND(getTemp3RealND());
  if (!mMatrixContainer.has(MI::kDtRho0Sgx)) SolverHipKernels::computeVelocityHomogeneousUniform<sd>(mMatrixContainer);
  else SolverHipKernels::computeVelocityHeterogeneous<sd>(mMatrixContainer); // per-voxel dt/rho0_sg (heterogeneous or non-uniform grid)
}

template<SD sd> void KSpaceFirstOrderSolver::computeVelocityGradient()
{ // :2126-2150
  if (mFused && mParameters.getNonUniformGridFlag() != 0)
  { // non-uniform grid: the gradient scaling sits between the gradient and the density update (:2145-2149), so the
    // gradients are produced as arrays by the fused FFT passes and the element-wise kernels of the reference follow
    const MatrixContainer& c = mMatrixContainer;
    kwCheck(kw_fused_velocity_gradient(mParameters.getHipParameters().getContext(), real(MI::kUxSgx).getDeviceData(),
                                       real(MI::kUySgy).getDeviceData(), real(MI::kUzSgz).getDeviceData(),
                                       real(MI::kDuxdx).getDeviceData(), real(MI::kDuydy).getDeviceData(),
                                       real(MI::kDuzdz).getDeviceData(), mKappaPadded,
                                       c.getMatrix<ComplexMatrix>(MI::kDdxKShiftNegR).getDeviceData(),
                                       c.getMatrix<ComplexMatrix>(MI::kDdyKShiftNeg).getDeviceData(),
                                       c.getMatrix<ComplexMatrix>(MI::kDdzKShiftNeg).getDeviceData(),
                                       mVelocityChained ? KW_FUSED_U_IN_SCRATCH : 0));
    SolverHipKernels::computeVelocityGradientShiftNonuniform<sd>(mMatrixContainer);
    return;
  }
  if (mFused) return; // folded into the density stage (kw_fused_density)
  getTempHipFftX().computeR2CFftND(real(MI::kUxSgx));
  getTempHipFftY().computeR2CFftND(real(MI::kUySgy));
  getTempHipFftZ().computeR2CFftND(real(MI::kUzSgz));
  SolverHipKernels::computeVelocityGradient<sd>(mMatrixContainer);
  getTempHipFftX().computeC2RFftND(real(MI::kDuxdx));
  getTempHipFftY().computeC2RFftND(real(MI::kDuydy));
  getTempHipFftZ().computeC2RFftND(real(MI::kDuzdz));
  if (mParameters.getNonUniformGridFlag() != 0) // :2145-2149
    SolverHipKernels::computeVelocityGradientShiftNonuniform<sd>(mMatrixContainer);
}

// fused: velocity gradient + density update (+ pressure terms when no pressure source sits between them this step)
void KSpaceFirstOrderSolver::fusedDensity(bool nonlinear)
{
  const MatrixContainer& c = mMatrixContainer;
  const bool absorbing     = mParameters.getAbsorbingFlag() != 0;
  const bool pSourceActive = mParameters.getPressureSourceFlag() > mParameters.getTimeIndex();
  mTermsFused              = absorbing && !pSourceActive;
  // lossless media: the equation of state (computePressure*'s lossless branch) is part of the density kernel, and the
  // spectrum of the new p is chained unless p is about to be overwritten by the initial pressure source (step 0)
  mPressureFused           = !absorbing && !pSourceActive;
  const bool chainP        = mPressureFused &&
                             !((mParameters.getTimeIndex() == 0) && (mParameters.getInitialPressureSourceFlag() == 1));
  const bool storeDu       = absorbing && pSourceActive; // the stand-alone terms kernel will need the gradients
  const int  terms         = mTermsFused ? (nonlinear ? 2 : 1) : (mPressureFused ? 3 : 0);
  const int  flags         = (mVelocityChained ? KW_FUSED_U_IN_SCRATCH : 0) |
                             ((mTermsFused || chainP) ? KW_FUSED_CHAIN_TERMS : 0);
  // aliasing of the temporaries as in :2184-2190 (nonlinear) / :2221-2225 (linear)
  float* t0 = getTemp1RealND().getDeviceData();
  float* t1 = getTemp2RealND().getDeviceData();
  float* t2 = getTemp3RealND().getDeviceData();
  if (mPressureFused)
  {
    t0 = getP().getDeviceData();
    t1 = const_cast<float*>(c.realDeviceOrNull(MI::kC2)); // input in this mode (see kw_fused_density)
    mPressureInScratch = chainP;
  }
  kwCheck(kw_fused_density(mParameters.getHipParameters().getContext(), nonlinear ? 1 : 0,
                           real(MI::kUxSgx).getDeviceData(), real(MI::kUySgy).getDeviceData(),
                           real(MI::kUzSgz).getDeviceData(), real(MI::kRhoX).getDeviceData(),
                           real(MI::kRhoY).getDeviceData(), real(MI::kRhoZ).getDeviceData(),
                           real(MI::kPmlX).getDeviceData(), real(MI::kPmlY).getDeviceData(),
                           real(MI::kPmlZ).getDeviceData(), c.realDeviceOrNull(MI::kRho0), mKappaPadded,
                           c.getMatrix<ComplexMatrix>(MI::kDdxKShiftNegR).getDeviceData(),
                           c.getMatrix<ComplexMatrix>(MI::kDdyKShiftNeg).getDeviceData(),
                           c.getMatrix<ComplexMatrix>(MI::kDdzKShiftNeg).getDeviceData(),
                           storeDu ? real(MI::kDuxdx).getDeviceData() : nullptr,
                           storeDu ? real(MI::kDuydy).getDeviceData() : nullptr,
                           storeDu ? real(MI::kDuzdz).getDeviceData() : nullptr, terms, c.realDeviceOrNull(MI::kBOnA),
                           t0, t1, t2, flags));
}

template<SD sd> void KSpaceFirstOrderSolver::computeDensityNonliner()
{
  if (mFused && mParameters.getNonUniformGridFlag() == 0) fusedDensity(true);
  else
  {
    mTermsFused = mPressureFused = false; // (non-uniform grid on the fused passes: element-wise kernels from here on)
    SolverHipKernels::computeDensityNonlinear<sd>(mMatrixContainer);
  }
}
template<SD sd> void KSpaceFirstOrderSolver::computeDensityLinear()
{
  if (mFused && mParameters.getNonUniformGridFlag() == 0) fusedDensity(false);
  else
  {
    mTermsFused = mPressureFused = false;
    SolverHipKernels::computeDensityLinear<sd>(mMatrixContainer);
  }
}

template<SD sd> void KSpaceFirstOrderSolver::computePressureNonlinear()
{ // :2180-2210
  if (mParameters.getAbsorbingFlag())
  {
    RealMatrix& densitySum          = getTemp1RealND();
    RealMatrix& nonlinearTerm       = getTemp2RealND();
    RealMatrix& velocityGradientSum = getTemp3RealND();
    RealMatrix& absorbTauTerm       = velocityGradientSum;
    RealMatrix& absorbEtaTerm       = densitySum;
    if (!(mFused && mTermsFused))
      SolverHipKernels::computePressureTermsNonlinear<sd>(densitySum, nonlinearTerm, velocityGradientSum, mMatrixContainer);
    if (mFused)
    {
      // the kernel that writes p also leaves its spectrum for the next step's velocity stage, unless p is about to be
      // overwritten by the initial pressure source (step 0)
      const bool chainP = !((mParameters.getTimeIndex() == 0) && (mParameters.getInitialPressureSourceFlag() == 1));
      kwCheck(kw_fused_absorption_pressure(mParameters.getHipParameters().getContext(), getP().getDeviceData(),
                                           velocityGradientSum.getDeviceData(), densitySum.getDeviceData(),
                                           nonlinearTerm.getDeviceData(), mNabla1Padded, mNabla2Padded,
                                           mMatrixContainer.realDeviceOrNull(MI::kC2),
                                           mMatrixContainer.realDeviceOrNull(MI::kAbsorbTau),
                                           mMatrixContainer.realDeviceOrNull(MI::kAbsorbEta),
                                           (mTermsFused ? KW_FUSED_TERMS_IN_SCRATCH : 0) | (chainP ? KW_FUSED_CHAIN_P : 0)));
      mPressureInScratch = chainP;
      return;
    }
    getTempHipFftX().computeR2CFftND(velocityGradientSum);
    getTempHipFftY().computeR2CFftND(densitySum);
    SolverHipKernels::computeAbsorbtionTerm(getTempHipFftX(), getTempHipFftY(), real(MI::kAbsorbNabla1), real(MI::kAbsorbNabla2));
    getTempHipFftX().computeC2RFftND(absorbTauTerm);
    getTempHipFftY().computeC2RFftND(absorbEtaTerm);
    SolverHipKernels::sumPressureTermsNonlinear(nonlinearTerm, absorbTauTerm, absorbEtaTerm, mMatrixContainer);
  }
  else if (!(mFused && mPressureFused))
  {
    SolverHipKernels::sumPressureNonlinearLossless<sd>(mMatrixContainer);
  }
}

template<SD sd> void KSpaceFirstOrderSolver::computePressureLinear()
{ // :2217-2245
  if (mParameters.getAbsorbingFlag())
  {
    RealMatrix& densitySum           = getTemp1RealND();
    RealMatrix& velocityGradientTerm = getTemp2RealND();
    RealMatrix& absorbTauTerm        = getTemp2RealND();
    RealMatrix& absorbEtaTerm        = getTemp3RealND();
    if (!(mFused && mTermsFused))
      SolverHipKernels::computePressureTermsLinear<sd>(densitySum, velocityGradientTerm, mMatrixContainer);
    if (mFused)
    {
      // the kernel that writes p also leaves its spectrum for the next step's velocity stage, unless p is about to be
      // overwritten by the initial pressure source (step 0)
      const bool chainP = !((mParameters.getTimeIndex() == 0) && (mParameters.getInitialPressureSourceFlag() == 1));
      kwCheck(kw_fused_absorption_pressure(mParameters.getHipParameters().getContext(), getP().getDeviceData(),
                                           velocityGradientTerm.getDeviceData(), densitySum.getDeviceData(),
                                           densitySum.getDeviceData(), mNabla1Padded, mNabla2Padded,
                                           mMatrixContainer.realDeviceOrNull(MI::kC2),
                                           mMatrixContainer.realDeviceOrNull(MI::kAbsorbTau),
                                           mMatrixContainer.realDeviceOrNull(MI::kAbsorbEta),
                                           (mTermsFused ? KW_FUSED_TERMS_IN_SCRATCH : 0) | (chainP ? KW_FUSED_CHAIN_P : 0)));
      mPressureInScratch = chainP;
      return;
    }
    getTempHipFftX().computeR2CFftND(velocityGradientTerm);
    getTempHipFftY().computeR2CFftND(densitySum);
    SolverHipKernels::computeAbsorbtionTerm(getTempHipFftX(), getTempHipFftY(), real(MI::kAbsorbNabla1), real(MI::kAbsorbNabla2));
    getTempHipFftX().computeC2RFftND(absorbTauTerm);
    getTempHipFftY().computeC2RFftND(absorbEtaTerm);
    SolverHipKernels::sumPressureTermsLinear(absorbTauTerm, absorbEtaTerm, densitySum, mMatrixContainer);
  }
  else if (!(mFused && mPressureFused))
  {
    SolverHipKernels::sumPressureLinearLossless<sd>(mMatrixContainer);
  }
}

void KSpaceFirstOrderSolver::addVelocitySource()
{ // :2252-2303
  const size_t timeIndex = mParameters.getTimeIndex();
  struct Comp { size_t flag; MI u; MI input; };
  const Comp comps[3] = {{mParameters.getVelocityXSourceFlag(), MI::kUxSgx, MI::kVelocityXSourceInput},
                         {mParameters.getVelocityYSourceFlag(), MI::kUySgy, MI::kVelocityYSourceInput},
                         {mParameters.getVelocityZSourceFlag(), MI::kUzSgz, MI::kVelocityZSourceInput}};
  for (const Comp& c : comps)
  {
    if (!(c.flag > timeIndex)) continue;
    if (mParameters.getVelocitySourceMode() != Parameters::SourceMode::kAdditive)
    {
      SolverHipKernels::addVelocitySource(real(c.u), real(c.input), index(MI::kVelocitySourceIndex));
    }
    else
    {
      RealMatrix& scaledSource = getTemp1RealND();
      scaleSource(scaledSource, real(c.input), index(MI::kVelocitySourceIndex), mParameters.getVelocitySourceMany());
      SolverHipKernels::addVelocityScaledSource(real(c.u), scaledSource);
    }
  }
}

template<SD sd> void KSpaceFirstOrderSolver::addPressureSource()
{ // :2310-2332
  if (mParameters.getPressureSourceFlag() > mParameters.getTimeIndex())
  {
    if (mParameters.getPressureSourceMode() != Parameters::SourceMode::kAdditive)
    {
      SolverHipKernels::addPressureSource<sd>(mMatrixContainer);
    }
    else
    {
      RealMatrix& scaledSource = getTemp1RealND();
      scaleSource(scaledSource, real(MI::kPressureSourceInput), index(MI::kPressureSourceIndex), mParameters.getPressureSourceMany());
      SolverHipKernels::addPressureScaledSource<sd>(mMatrixContainer, scaledSource);
    }
  }
}

void KSpaceFirstOrderSolver::scaleSource(RealMatrix& scaledSource, const RealMatrix& sourceInput,
                                         const IndexMatrix& sourceIndex, const size_t manyFlag)
{ // :2339-2352
  scaledSource.zeroDeviceMatrix();
  SolverHipKernels::insertSourceIntoScalingMatrix(scaledSource, sourceInput, sourceIndex, manyFlag);
  if (mFused)
  {
    kwCheck(kw_fused_scale_source(mParameters.getHipParameters().getContext(), scaledSource.getDeviceData(), mSourceKappaPadded));
    return;
  }
  getTempHipFftX().computeR2CFftND(scaledSource);
  SolverHipKernels::computeSourceGradient(getTempHipFftX(), real(MI::kSourceKappa));
  getTempHipFftX().computeC2RFftND(scaledSource);
}

template<SD sd> void KSpaceFirstOrderSolver::addInitialPressureSource()
{ // :2359-2396
  SolverHipKernels::addInitialPressureSource<sd>(mMatrixContainer);
  if (mFused)
  {
    const MatrixContainer& c = mMatrixContainer;
    kwCheck(kw_fused_initial_velocity(mParameters.getHipParameters().getContext(), getP().getDeviceData(),
                                      real(MI::kUxSgx).getDeviceData(), real(MI::kUySgy).getDeviceData(),
                                      real(MI::kUzSgz).getDeviceData(), c.realDeviceOrNull(MI::kDtRho0Sgx),
                                      c.realDeviceOrNull(MI::kDtRho0Sgy), c.realDeviceOrNull(MI::kDtRho0Sgz),
                                      mKappaPadded, c.getMatrix<ComplexMatrix>(MI::kDdxKShiftPosR).getDeviceData(),
                                      c.getMatrix<ComplexMatrix>(MI::kDdyKShiftPos).getDeviceData(),
                                      c.getMatrix<ComplexMatrix>(MI::kDdzKShiftPos).getDeviceData()));
    return;
  }
  getTempHipFftX().computeR2CFftND(getP());
  SolverHipKernels::computePressureGradient<sd>(mMatrixContainer);
  getTempHipFftX().computeC2RFftND(real(MI::kUxSgx));
  getTempHipFftY().computeC2RFftND(real(MI::kUySgy));
  getTempHipFftZ().computeC2RFftND(real(MI::kUzSgz));
  if (!mMatrixContainer.has(MI::kDtRho0Sgx)) SolverHipKernels::computeInitialVelocityHomogeneousUniform<sd>(mMatrixContainer);
  else SolverHipKernels::computeInitialVelocityHeterogeneous<sd>(mMatrixContainer);
}

template<SD sd> void KSpaceFirstOrderSolver::computeShiftedVelocity()
{ // :2714-2735
  if (mFused && mShiftFilter[0] != nullptr)
  { // fast path: forward transform, shift and inverse transform of an axis in one kernel (kw_fused_shift_velocity)
    kw_ctx* ctx = mParameters.getHipParameters().getContext();
    kwCheck(kw_fused_shift_velocity(ctx, 0, real(MI::kUxSgx).getDeviceData(), real(MI::kUxShifted).getDeviceData(), mShiftFilter[0]));
    kwCheck(kw_fused_shift_velocity(ctx, 1, real(MI::kUySgy).getDeviceData(), real(MI::kUyShifted).getDeviceData(), mShiftFilter[1]));
    if (mParameters.isSimulation3D())
      kwCheck(kw_fused_shift_velocity(ctx, 2, real(MI::kUzSgz).getDeviceData(), real(MI::kUzShifted).getDeviceData(), mShiftFilter[2]));
    return;
  }
  HipFftComplexMatrix& tempShift = fft(MI::kTempHipFftShift);
  tempShift.computeR2CFft1DX(real(MI::kUxSgx));
  SolverHipKernels::computeVelocityShiftInX(tempShift, mMatrixContainer.getMatrix<ComplexMatrix>(MI::kXShiftNegR));
  tempShift.computeC2RFft1DX(real(MI::kUxShifted));
  tempShift.computeR2CFft1DY(real(MI::kUySgy));
  SolverHipKernels::computeVelocityShiftInY(tempShift, mMatrixContainer.getMatrix<ComplexMatrix>(MI::kYShiftNegR));
  tempShift.computeC2RFft1DY(real(MI::kUyShifted));
  if (!mParameters.isSimulation3D()) return; // SD::k2D instantiation: x and y only (:2714-2735)
  tempShift.computeR2CFft1DZ(real(MI::kUzSgz));
  SolverHipKernels::computeVelocityShiftInZ(tempShift, mMatrixContainer.getMatrix<ComplexMatrix>(MI::kZShiftNegR));
  tempShift.computeC2RFft1DZ(real(MI::kUzShifted));
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused pipeline set-up: the reduced operators are imported once into the pipeline's padded row layout
// ---------------------------------------------------------------------------------------------------------------------
float* KSpaceFirstOrderSolver::importPadded(MI idx)
{
  kw_ctx* ctx = mParameters.getHipParameters().getContext();
  size_t  n   = 0;
  kwCheck(kw_fused_reduced_elems(ctx, &n));
  void* d = nullptr;
  kwCheck(kw_malloc(ctx, n * sizeof(float), &d));
  kwCheck(kw_fused_import_reduced(ctx, static_cast<float*>(d), real(idx).getDeviceData()));
  return static_cast<float*>(d);
}

void KSpaceFirstOrderSolver::initializeFusedPipeline()
{
  // step graphs (Options::stepGraph): measured 3-5 % SLOWER than eager launches at 64^3 and 128^3 on this stack (the small
  // grids are bound by the device-side kernel boundaries, ~7 us per kernel, not by host launch cost) — off by default
  mUseStepGraph = mParameters.getOptions().stepGraph;
  if (mParameters.isSlabDecomposed()) mUseStepGraph = false; // the exchange callbacks cannot be recorded
  mKappaPadded = importPadded(MI::kKappa);
  if (mMatrixContainer.has(MI::kAbsorbNabla1))
  {
    mNabla1Padded = importPadded(MI::kAbsorbNabla1);
    mNabla2Padded = importPadded(MI::kAbsorbNabla2);
  }
  if (mMatrixContainer.has(MI::kSourceKappa)) mSourceKappaPadded = importPadded(MI::kSourceKappa);
  if (mParameters.needsShiftedVelocity())
  { // filters of the one-kernel-per-axis shift: the half-length shift vectors of the input file extended to full length
    // the way R2C -> multiply -> C2R acts on a real line (imaginary parts of the DC and Nyquist bins are dropped), with
    // the 1/N of the transform pair folded in (the reference multiplies by it in computeVelocityShiftIn*, .cu:2617-2710)
    kw_ctx* ctx = mParameters.getHipParameters().getContext();
    const DimensionSizes dims = mParameters.getGlobalDimensionSizes(); // z lines have the global length in slab mode
    const size_t n[3]   = {dims.nx, dims.ny, dims.nz};
    const MI     idx[3] = {MI::kXShiftNegR, MI::kYShiftNegR, MI::kZShiftNegR};
    const int    axes   = mParameters.isSimulation3D() ? 3 : 2;
    for (int a = 0; a < axes; a++)
    {
      const float* half = mMatrixContainer.getMatrix<ComplexMatrix>(idx[a]).getHostData(); // (re, im) pairs
      const float  divider = 1.0f / static_cast<float>(n[a]);
      std::vector<float> full(2 * n[a], 0.0f);
      full[0] = half[0] * divider;
      for (size_t k = 1; k < (n[a] + 1) / 2; k++)
      {
        full[2 * k]     = full[2 * (n[a] - k)] = half[2 * k] * divider;
        full[2 * k + 1] = half[2 * k + 1] * divider;
        full[2 * (n[a] - k) + 1] = -(half[2 * k + 1] * divider);
      }
      if (n[a] % 2 == 0) full[n[a]] = half[n[a]] * divider; // index 2 * (n/2): real part of the Nyquist bin
      void* d = nullptr;
      kwCheck(kw_malloc(ctx, full.size() * sizeof(float), &d));
      kwCheck(kw_memcpy_h2d(ctx, d, full.data(), full.size() * sizeof(float)));
      mShiftFilter[a] = static_cast<float*>(d);
    }
  }
}

void KSpaceFirstOrderSolver::releaseFusedPipeline()
{
  kw_ctx* ctx = mParameters.getHipParameters().getContext();
  if (!ctx) return;
  float** bufs[] = { &mKappaPadded, &mNabla1Padded, &mNabla2Padded, &mSourceKappaPadded, &mShiftFilter[0], &mShiftFilter[1],
                     &mShiftFilter[2] };
  for (float** b : bufs)
  {
    if (*b) kw_free(ctx, *b);
    *b = nullptr;
  }
  if (mStepGraph) (void)kw_graph_destroy(ctx, mStepGraph);
  mStepGraph = nullptr;
  if (mFused) kw_fused_destroy(ctx);
  mFused = false;
  if (mOwnsComm) (void)kw_comm_destroy(ctx);
  mOwnsComm = false;
}

// ---------------------------------------------------------------------------------------------------------------------
// Host generators.  The operators are computed, not read from the file, so their fp32
// rounding has to be the reference's (KSpaceFirstOrderSolver.cpp:2404-2703): per bin
//     s = tx[x] + (tz[z] + ty[y]),   t_axis[i] = ((0.5 - |0.5 - i * (1/N)|)^2) * (1/d^2)
// in exactly that association.  Each t depends on one coordinate only: three short tables are built once (the y table
// starts at this rank's first ky in slab mode) and one sweep over the bins feeds whichever operators are wanted.
// ---------------------------------------------------------------------------------------------------------------------
namespace
{
struct SpectralBins
{
  std::vector<float> tx, ty, tz; // squared normalised frequency over squared spacing, per axis
  size_t             nxr = 0, nyl = 0, nzg = 0;

  static std::vector<float> table(size_t count, size_t first, size_t n, float spacing, bool flat)
  {
    std::vector<float> t(count);
    const float nRec  = 1.0f / static_cast<float>(n);
    const float d2Rec = flat ? 0.0f : 1.0f / (spacing * spacing); // an axis a 2-D grid does not have contributes nothing
    for (size_t i = 0; i < count; i++)
    {
      const float f = 0.5f - std::fabs(0.5f - static_cast<float>(first + i) * nRec);
      t[i]          = (f * f) * d2Rec;
    }
    return t;
  }

  explicit SpectralBins(const Parameters& par)
  {
    const DimensionSizes full = par.getGlobalDimensionSizes();
    nxr = par.getReducedDimensionSizes().nx;
    nzg = full.nz;
    // one GPU: [nz][ny][nxr]; Z-slab mode: this rank's spectra live transposed, [nz_global][ny / ranks][nxr]
    nyl = full.ny / par.getSlabRanks();
    tx  = table(nxr, 0, full.nx, par.getDx(), false);
    ty  = table(nyl, par.getSlabRank() * nyl, full.ny, par.getDy(), false);
    tz  = table(nzg, 0, full.nz, par.isSimulation3D() ? par.getDz() : 1.0f, !par.isSimulation3D());
  }

  // f(i, r) for every bin i of the reduced grid with r = sqrt(s)
  template<class F> void sweep(F&& f) const
  {
#pragma omp parallel for schedule(static) collapse(2)
    for (size_t z = 0; z < nzg; z++)
      for (size_t y = 0; y < nyl; y++)
      {
        const float tzy = tz[z] + ty[y];
        const size_t row = (z * nyl + y) * nxr;
        for (size_t x = 0; x < nxr; x++) f(row + x, std::sqrt(tx[x] + tzy));
      }
  }
};

inline float sincOrOne(float a) { return (a == 0.0f) ? 1.0f : std::sin(a) / a; }
inline float finiteOrZero(float v) { return (v == std::numeric_limits<float>::infinity()) ? 0.0f : v; }
} // namespace

void KSpaceFirstOrderSolver::generateKappa()
{ // kappa = sinc(c_ref * dt * pi * |f|)
  const float scale = mParameters.getCRef() * mParameters.getDt() * static_cast<float>(M_PI);
  float* kappa = real(MI::kKappa).getHostData();
  SpectralBins(mParameters).sweep([=](size_t i, float r) { kappa[i] = sincOrOne(scale * r); });
}

void KSpaceFirstOrderSolver::generateSourceKappa()
{ // source_kappa = cos(c_ref * dt * pi * |f|)
  const float scale = mParameters.getCRef() * mParameters.getDt() * static_cast<float>(M_PI);
  float* sourceKappa = real(MI::kSourceKappa).getHostData();
  SpectralBins(mParameters).sweep([=](size_t i, float r) { sourceKappa[i] = std::cos(scale * r); });
}

void KSpaceFirstOrderSolver::generateKappaAndNablas()
{ // k = 2 pi |f|;  kappa = sinc(c_ref * dt / 2 * k),  nabla1 = k^(y - 2),  nabla2 = k^(y - 1), infinities (k = 0) -> 0
  const float halfStep = mParameters.getCRef() * mParameters.getDt() * 0.5f;
  const float twoPi    = static_cast<float>(M_PI) * 2.0f;
  const float power    = mParameters.getAlphaPower();
  float* kappa  = real(MI::kKappa).getHostData();
  float* nabla1 = real(MI::kAbsorbNabla1).getHostData();
  float* nabla2 = real(MI::kAbsorbNabla2).getHostData();
  SpectralBins(mParameters).sweep([=](size_t i, float r) {
    const float k = twoPi * r;
    kappa[i]      = sincOrOne(halfStep * k);
    nabla1[i]     = finiteOrZero(std::pow(k, power - 2.0f));
    nabla2[i]     = finiteOrZero(std::pow(k, power - 1.0f));
  });
}

void KSpaceFirstOrderSolver::generateTauAndEta()
{ // tau = -2 a c0^(y-1),  eta = 2 a c0^y tan(pi y / 2),  a = alpha_coeff * 100 (1e-6 / 2 pi)^y / (20 log10 e)  [Np]
  const float power   = mParameters.getAlphaPower();
  const float tanTerm = std::tan(static_cast<float>(M_PI_2) * power);
  const float neper   = (100.0f * std::pow(1.0e-6f / (2.0f * static_cast<float>(M_PI)), power)) / (20.0f * static_cast<float>(M_LOG10E));
  const bool  alphaIsScalar = mParameters.getAlphaCoeffScalarFlag(), c0IsScalar = mParameters.getC0ScalarFlag();
  if (alphaIsScalar && c0IsScalar)
  {
    const float a2 = 2.0f * mParameters.getAlphaCoeffScalar() * neper;
    mParameters.setAbsorbTauScalar((-a2) * std::pow(mParameters.getC0Scalar(), power - 1));
    mParameters.setAbsorbEtaScalar(a2 * std::pow(mParameters.getC0Scalar(), power) * tanTerm);
    return;
  }
  // per voxel as soon as either operand is an array; alpha_coeff was loaded into Temp1, the c2 matrix still holds c0
  const float* alpha = alphaIsScalar ? nullptr : getTemp1RealND().getHostData();
  const float* c0    = c0IsScalar ? nullptr : real(MI::kC2).getHostData();
  const float  alphaScalar = alphaIsScalar ? mParameters.getAlphaCoeffScalar() : 0.0f;
  const float  c0Scalar    = c0IsScalar ? mParameters.getC0Scalar() : 0.0f;
  float* tau = real(MI::kAbsorbTau).getHostData();
  float* eta = real(MI::kAbsorbEta).getHostData();
  const size_t n = mParameters.getFullDimensionSizes().nElements();
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++)
  {
    const float a2 = 2.0f * neper * (alpha ? alpha[i] : alphaScalar);
    const float c  = c0 ? c0[i] : c0Scalar;
    tau[i] = (-a2) * std::pow(c, power - 1.0f);
    eta[i] = a2 * std::pow(c, power) * tanTerm;
  }
}

void KSpaceFirstOrderSolver::computeC2()
{ // c0 -> c0^2 in place (heterogeneous sound speed only; the scalar is squared in Parameters)
  if (mParameters.getC0ScalarFlag()) return;
  RealMatrix& c = real(MI::kC2);
  float* v = c.getHostData();
  const size_t n = c.size();
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) v[i] *= v[i];
}

// dt/rho0_sg on a non-uniform grid (KSpaceFirstOrderSolver.cpp:2650-2685 for heterogeneous density).  For a homogeneous
// density the reference multiplies dt/rho0_sg * d?ud?n_sg? inside its own kernel variant (SolverCudaKernels.cu:372-410,
// 1061-1083); here the product is stored per voxel and the heterogeneous kernels are used.
void KSpaceFirstOrderSolver::generateInitialDenisty()
{
  const DimensionSizes d = mParameters.getFullDimensionSizes();
  float* sgx = real(MI::kDtRho0Sgx).getHostData();
  float* sgy = real(MI::kDtRho0Sgy).getHostData();
  float* sgz = real(MI::kDtRho0Sgz).getHostData();
  const float* nx = real(MI::kDxudxnSgx).getHostData();
  const float* ny = real(MI::kDyudynSgy).getHostData();
  const float* nz = real(MI::kDzudznSgz).getHostData();
  const float dt = mParameters.getDt();
  const bool scalar = mParameters.getRho0ScalarFlag();
  const float sx = mParameters.getDtRho0SgxScalar(), sy = mParameters.getDtRho0SgyScalar(), sz = mParameters.getDtRho0SgzScalar();
#pragma omp parallel for schedule(static)
  for (size_t z = 0; z < d.nz; z++)
    for (size_t y = 0; y < d.ny; y++)
      for (size_t x = 0; x < d.nx; x++)
      {
        const size_t i = (z * d.ny + y) * d.nx + x;
        if (scalar)
        {
          sgx[i] = sx * nx[x];
          sgy[i] = sy * ny[y];
          sgz[i] = sz * nz[z];
        }
        else
        {
          sgx[i] = (dt * nx[x]) / sgx[i];
          sgy[i] = (dt * ny[y]) / sgy[i];
          sgz[i] = (dt * nz[z]) / sgz[i];
        }
      }
}
