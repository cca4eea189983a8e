// OutputStreams.cpp — see OutputStreams.h.
#include "OutputStreams.h"
#include <cstdlib>
#include <new>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <stdexcept>

#include "CompressHelper.h"
#include "HipError.h"
#include "MatrixNames.h"
#include "Parameters.h"

static kw_ctx* ctx() { return Parameters::getInstance().getHipParameters().getContext(); }

namespace OutputStreamsHipKernels
{
void sampleIndex(ReduceOperator op, float* buf, const float* src, const size_t* mask, size_t n)
{
  kwCheck(kw_sample_index(ctx(), static_cast<kw_reduce_op>(op), buf, src, (const uint64_t*)mask, n));
}
void sampleIndexMulti(int nOps, const ReduceOperator* ops, float* const* bufs, const float* src, const size_t* mask, size_t n)
{
  kw_reduce_op kops[4];
  for (int o = 0; o < nOps && o < 4; o++) kops[o] = static_cast<kw_reduce_op>(ops[o]);
  kwCheck(kw_sample_index_multi(ctx(), nOps, kops, bufs, src, (const uint64_t*)mask, n));
}
void sampleCuboid(ReduceOperator op, float* buf, const float* src, const DimensionSizes& tl, const DimensionSizes& br,
                  const DimensionSizes& size, size_t n)
{
  const uint32_t a[3] = {(uint32_t)tl.nx, (uint32_t)tl.ny, (uint32_t)tl.nz};
  const uint32_t b[3] = {(uint32_t)br.nx, (uint32_t)br.ny, (uint32_t)br.nz};
  const uint32_t s[3] = {(uint32_t)size.nx, (uint32_t)size.ny, (uint32_t)size.nz};
  kwCheck(kw_sample_cuboid(ctx(), static_cast<kw_reduce_op>(op), buf, src, a, b, s, n));
}
void sampleAll(ReduceOperator op, float* buf, const float* src, size_t n)
{
  kwCheck(kw_sample_all(ctx(), static_cast<kw_reduce_op>(op), buf, src, n));
}
void postProcessingRms(float* buf, float coeff, size_t n) { kwCheck(kw_post_processing_rms(ctx(), buf, coeff, n)); }
} // namespace OutputStreamsHipKernels

// ---- BaseOutputStream -----------------------------------------------------------------------------------------------
BaseOutputStream::~BaseOutputStream() { freeMemory(); }

OutputStreamsHipKernels::ReduceOperator BaseOutputStream::kernelOp() const
{
  using K = OutputStreamsHipKernels::ReduceOperator;
  switch (mReduceOp)
  {
    case ReduceOperator::kRms: return K::kRms;
    case ReduceOperator::kMax: return K::kMax;
    case ReduceOperator::kMin: return K::kMin;
    default: return K::kNone;
  }
}

void BaseOutputStream::allocateMemory()
{
  void* d = nullptr;
  kwCheck(kw_malloc(ctx(), mSize * sizeof(float), &d));
  mDeviceBuffer = static_cast<float*>(d);
  // initial values per reduce operator (BaseOutputStream.cpp:271-367)
  std::vector<float> init(mSize, 0.0f);
  if (mReduceOp == ReduceOperator::kMax) init.assign(mSize, -1 * std::numeric_limits<float>::max());
  if (mReduceOp == ReduceOperator::kMin) init.assign(mSize, std::numeric_limits<float>::max());
  kwCheck(kw_memcpy_h2d(ctx(), mDeviceBuffer, init.data(), mSize * sizeof(float)));
  if (mReduceOp == ReduceOperator::kNone)
  {
    // the whole series is known in advance (Nt - s rows): growing the vector step by step would re-copy it several
    // times and stall the launching thread for milliseconds at a time, long enough for the GPU queue to run dry
    const Parameters& params = Parameters::getInstance();
    if (params.getNt() > params.getSamplingStartTimeIndex())
    {
      try { mDataset.reserve((params.getNt() - params.getSamplingStartTimeIndex()) * mSize); }
      catch (const std::bad_alloc&) {} // a series this long is better compressed or checkpointed; fall back to growing
    }
    mDeviceRaw[0] = mDeviceBuffer;
    void* d2 = nullptr;
    kwCheck(kw_malloc(ctx(), mSize * sizeof(float), &d2));
    mDeviceRaw[1] = static_cast<float*>(d2);
    for (int b = 0; b < 2; b++)
    {
      void* h = nullptr;
      kwCheck(kw_host_alloc(ctx(), mSize * sizeof(float), &h));
      mPinned[b] = static_cast<float*>(h);
      kwCheck(kw_event_create(ctx(), &mEvent[b]));
    }
  }
}

void BaseOutputStream::freeMemory()
{
  if (!ctx()) return;
  if (mDeviceBuffer) kw_free(ctx(), mDeviceBuffer);
  if (mDeviceRaw[1]) kw_free(ctx(), mDeviceRaw[1]);
  mDeviceBuffer = nullptr;
  mDeviceRaw[0] = mDeviceRaw[1] = nullptr;
  for (int b = 0; b < 2; b++)
  {
    if (mPinned[b]) kw_host_free(ctx(), mPinned[b]);
    if (mEvent[b]) kw_event_destroy(ctx(), mEvent[b]);
    mPinned[b] = nullptr;
    mEvent[b]  = nullptr;
  }
}

void BaseOutputStream::storeRow(const float* row)
{
  if (mSink) mSink->append(row, mSize);
  else mDataset.insert(mDataset.end(), row, row + mSize);
}

void BaseOutputStream::loadSeries()
{
  if (!mSink || !isSeries()) return;
  mSink->flush();
  mSink->read(mDataset, mFlushedSteps);
}

void BaseOutputStream::adoptStoredSeries(size_t rows)
{ // --post: the series was written by an earlier run; this stream only reads it back
  if (!mSink || !isSeries()) throw std::runtime_error("stream " + mName + " has no stored series to adopt");
  mSink->setRows(rows);
  mSampledSteps = mFlushedSteps = rows;
}

void BaseOutputStream::copyAggregateFromDevice()
{
  mDataset.resize(mSize);
  kwCheck(kw_memcpy_d2h(ctx(), mDataset.data(), mDeviceBuffer, mSize * sizeof(float)));
}

void BaseOutputStream::postProcess()
{
  if (mReduceOp == ReduceOperator::kRms)
  { // BaseOutputStream.cpp:172-178
    const Parameters& p = Parameters::getInstance();
    const float scalingCoeff = 1.0f / (p.getNt() - p.getSamplingStartTimeIndex());
    OutputStreamsHipKernels::postProcessingRms(mDeviceBuffer, scalingCoeff, mSize);
  }
  if (mReduceOp != ReduceOperator::kNone && mReduceOp != ReduceOperator::kC) copyAggregateFromDevice();
}

void BaseOutputStream::checkpointState(std::vector<float>& state, size_t& sampledSteps)
{
  if (mReduceOp == ReduceOperator::kC || mReduceOp == ReduceOperator::kIAvgC)
    throw std::runtime_error("checkpointing of compression streams (" + mName + ") is not implemented");
  if (mReduceOp == ReduceOperator::kNone)
  {
    while (mFlushedSteps < mSampledSteps) flushRaw(); // the step sampled last is still in its pinned buffer
    if (mSink) { mSink->flush(); state.clear(); }     // the series is in the output file (as in the reference)
    else state = mDataset;
  }
  else
  { // accumulator as it stands (RMS: sum of squares, scaled only in postProcess)
    state.resize(mSize);
    kwCheck(kw_memcpy_d2h(ctx(), state.data(), mDeviceBuffer, mSize * sizeof(float)));
  }
  sampledSteps = mSampledSteps;
}

void BaseOutputStream::restoreState(const float* state, size_t n, size_t sampledSteps)
{
  if (mReduceOp == ReduceOperator::kC || mReduceOp == ReduceOperator::kIAvgC)
    throw std::runtime_error("checkpointing of compression streams (" + mName + ") is not implemented");
  if (mReduceOp == ReduceOperator::kNone)
  {
    if (mSink)
    { // rows 0 .. sampledSteps-1 are in the re-opened output file
      if (n != 0) throw std::invalid_argument("checkpoint of stream " + mName + " carries a series but the output file is streamed");
      mSink->setRows(sampledSteps);
    }
    else
    {
      if (n != sampledSteps * mSize) throw std::invalid_argument("checkpoint of stream " + mName + " has the wrong size");
      mDataset.assign(state, state + n);
    }
    mSampledSteps = mFlushedSteps = sampledSteps;
  }
  else
  {
    if (n != mSize) throw std::invalid_argument("checkpoint of stream " + mName + " has the wrong size");
    kwCheck(kw_memcpy_h2d(ctx(), mDeviceBuffer, state, mSize * sizeof(float)));
    mSampledSteps = sampledSteps;
  }
}

// ---- raw helpers ----------------------------------------------------------------------------------------------------
// (the reference's sampling kernel writes zero-copy mapped host buffers itself, BaseOutputStream.cpp:369-388; measured
// equal to this staged copy on the copy stream — both ~10 us per step at 256^3 with 65 536 points)
static void rawSampleTail(kw_ctx* c, float* dev, float* pinned, void* event, size_t n)
{
  kwCheck(kw_memcpy_d2h_overlapped(c, pinned, dev, n * sizeof(float), event));
}

// ---- IndexOutputStream ----------------------------------------------------------------------------------------------
void IndexOutputStream::create()
{
  mSize = mSensorMask.size();
  allocateMemory();
}
void IndexOutputStream::sample()
{
  OutputStreamsHipKernels::sampleIndex(kernelOp(), sampleTarget(), mSourceMatrix.getDeviceData(), mSensorMask.getDeviceData(), mSize);
  sampleDone();
}
float* IndexOutputStream::sampleTarget()
{
  if (mReduceOp != ReduceOperator::kNone) return mDeviceBuffer;
  return mDeviceRaw[mSampledSteps & 1];
}
void IndexOutputStream::sampleDone()
{
  const int b = mSampledSteps & 1;
  if (mReduceOp == ReduceOperator::kNone) rawSampleTail(ctx(), mDeviceRaw[b], mPinned[b], mEvent[b], mSize);
  mSampledSteps++;
}
void IndexOutputStream::flushRaw()
{
  if (mReduceOp != ReduceOperator::kNone || mFlushedSteps >= mSampledSteps) return;
  const int b = mFlushedSteps & 1;
  kwCheck(kw_event_synchronize(ctx(), mEvent[b])); // IndexOutputStream.cpp:354
  storeRow(mPinned[b]);
  mFlushedSteps++;
}

// ---- CompressedIndexOutputStream ------------------------------------------------------------------------------------
CompressedIndexOutputStream::~CompressedIndexOutputStream()
{
  if (!ctx()) return;
  float* bufs[] = { mC1, mC2, mBE, mBE_1 };
  if (mC2 == mC1) bufs[1] = nullptr;
  for (float* b : bufs)
    if (b) kw_free(ctx(), b);
}
void CompressedIndexOutputStream::create()
{
  const CompressHelper& ch = CompressHelper::getInstance();
  m40bit = Parameters::getInstance().get40bitCompressionFlag();
  // floats per frame: a complex coefficient is 2 floats, or 5 bytes = 1.25 floats (IndexOutputStream.cpp:90-93)
  mSize = m40bit ? static_cast<size_t>(std::ceil(mSensorMask.size() * 1.25f)) * ch.getHarmonics()
                 : mSensorMask.size() * ch.getHarmonics() * 2;
  auto dalloc = [&](size_t floats) {
    void* d = nullptr;
    kwCheck(kw_malloc(ctx(), floats * sizeof(float), &d));
    kwCheck(kw_memset(ctx(), d, 0, floats * sizeof(float)));
    return static_cast<float*>(d);
  };
  mC1 = dalloc(mSize);
  mC2 = Parameters::getInstance().getNoCompressionOverlapFlag() ? mC1 : dalloc(mSize); // BaseOutputStream.cpp:246-256
  const size_t bFloats = ch.getHarmonics() * ch.getBSize() * 2;
  mBE   = dalloc(bFloats);
  mBE_1 = dalloc(bFloats);
  kwCheck(kw_memcpy_h2d(ctx(), mBE, mShifted ? ch.getBEShifted() : ch.getBE(), bFloats * sizeof(float)));
  kwCheck(kw_memcpy_h2d(ctx(), mBE_1, mShifted ? ch.getBE_1Shifted() : ch.getBE_1(), bFloats * sizeof(float)));
  mFrameHost.resize(mSize);
}
void CompressedIndexOutputStream::sample()
{ // IndexOutputStream.cpp:380-389 (flags) and :403-452 (correlation), at sampling time
  const CompressHelper& ch = CompressHelper::getInstance();
  const Parameters& params = Parameters::getInstance();
  const size_t stepLocal   = mSampledSteps % (ch.getBSize() - 1);
  mSavingFlag              = ((stepLocal + 1) % ch.getOSize() == 0);
  const bool oddFrameFlag  = ((mCompressedTimeStep + 1) % 2 == 0);
  const bool mirror = (mCompressedTimeStep == 0 && mSavingFlag && !params.getNoCompressionOverlapFlag());
  if (m40bit)
    kwCheck(kw_sample_index_compress_40b(ctx(), mC1, mC2, mSourceMatrix.getDeviceData(),
                                         (const uint64_t*)mSensorMask.getDeviceData(), mSensorMask.size(),
                                         static_cast<uint32_t>(ch.getHarmonics()), mBE, mBE_1,
                                         static_cast<uint32_t>(ch.getBSize()), static_cast<uint32_t>(stepLocal), mirror ? 1 : 0,
                                         params.getNoCompressionOverlapFlag() ? 1 : 0, maxExp()));
  else
    kwCheck(kw_sample_index_compress(ctx(), mC1, mC2, mSourceMatrix.getDeviceData(),
                                     (const uint64_t*)mSensorMask.getDeviceData(), mSensorMask.size(),
                                     static_cast<uint32_t>(ch.getHarmonics()), mBE, mBE_1,
                                     static_cast<uint32_t>(ch.getBSize()), static_cast<uint32_t>(stepLocal), mirror ? 1 : 0));
  const size_t steps  = params.getNt() - params.getSamplingStartTimeIndex();
  const bool lastStep = ((steps - mSampledSteps == 1) && steps <= ch.getOSize());
  mCurrent = (mSavingFlag || lastStep) ? (oddFrameFlag ? mC1 : mC2) : nullptr; // :456-459
  mSampledSteps++;
}
int CompressedIndexOutputStream::maxExp() const
{ // BaseOutputStream.cpp:68-83: the streams on the time-shifted basis (u_non_staggered_c) use the velocity bias
  return mShifted ? CompressHelper::kMaxExpU : CompressHelper::kMaxExpP;
}

void CompressedIndexOutputStream::postSample2()
{
  if (mCurrent == nullptr) return;
  kwCheck(kw_memcpy_d2h(ctx(), mFrameHost.data(), mCurrent, mSize * sizeof(float)));
  storeRow(mFrameHost.data());
  mFlushedSteps++;
  mCompressedTimeStep++;
  if (mSavingFlag) kwCheck(kw_memset(ctx(), mCurrent, 0, mSize * sizeof(float))); // BaseOutputStream.cpp:117-132
  mCurrent = nullptr;
}

void CompressedIndexOutputStream::checkpointState(std::vector<float>& state, size_t& sampledSteps)
{ // taken between steps: the frame finished by the last sampled step has already been emitted (postSample2)
  if (mSink) { mSink->flush(); state.clear(); } // frames so far are in the output file
  else state = mDataset;
  const size_t frames = state.size();
  const int    nacc   = (mC2 == mC1) ? 1 : 2;
  state.resize(frames + nacc * mSize);
  kwCheck(kw_memcpy_d2h(ctx(), state.data() + frames, mC1, mSize * sizeof(float)));
  if (nacc == 2) kwCheck(kw_memcpy_d2h(ctx(), state.data() + frames + mSize, mC2, mSize * sizeof(float)));
  sampledSteps = mSampledSteps;
}
void CompressedIndexOutputStream::accumulators(std::vector<float>& c1, std::vector<float>& c2)
{
  if (mSink) mSink->flush();
  c1.resize(mSize);
  c2.resize(mSize);
  kwCheck(kw_memcpy_d2h(ctx(), c1.data(), mC1, mSize * sizeof(float)));
  kwCheck(kw_memcpy_d2h(ctx(), c2.data(), mC2, mSize * sizeof(float)));
}
void CompressedIndexOutputStream::restoreAccumulators(const float* c1, const float* c2, size_t n, size_t sampledSteps)
{
  if (n != mSize) throw std::invalid_argument("checkpoint of stream " + mName + " has the wrong size");
  mCompressedTimeStep = mFlushedSteps = sampledSteps / CompressHelper::getInstance().getOSize(); // IndexOutputStream.cpp:204-209
  if (mSink) mSink->setRows(mCompressedTimeStep);
  else mDataset.clear();
  kwCheck(kw_memcpy_h2d(ctx(), mC1, c1, mSize * sizeof(float)));
  if (mC2 != mC1) kwCheck(kw_memcpy_h2d(ctx(), mC2, c2, mSize * sizeof(float)));
  mSampledSteps = sampledSteps;
  mCurrent      = nullptr;
}
void CompressedIndexOutputStream::restoreState(const float* state, size_t n, size_t sampledSteps)
{
  const size_t nacc = (mC2 == mC1) ? 1 : 2;
  if (n < nacc * mSize || (n - nacc * mSize) % mSize != 0)
    throw std::invalid_argument("checkpoint of stream " + mName + " has the wrong size");
  const size_t frames = n - nacc * mSize;
  if (mSink)
  { // the number of frames in the re-opened output file follows from the sampled steps (one per oSize steps)
    if (frames != 0) throw std::invalid_argument("checkpoint of stream " + mName + " carries frames but the output file is streamed");
    mCompressedTimeStep = mFlushedSteps = sampledSteps / CompressHelper::getInstance().getOSize();
    mSink->setRows(mCompressedTimeStep);
  }
  else
  {
    mDataset.assign(state, state + frames);
    mCompressedTimeStep = mFlushedSteps = frames / mSize;
  }
  kwCheck(kw_memcpy_h2d(ctx(), mC1, state + frames, mSize * sizeof(float)));
  if (nacc == 2) kwCheck(kw_memcpy_h2d(ctx(), mC2, state + frames + mSize, mSize * sizeof(float)));
  mSampledSteps = sampledSteps;
  mCurrent      = nullptr;
}

// ---- IntensityAvgCOutputStream --------------------------------------------------------------------------------------
void IntensityAvgCOutputStream::checkpointState(std::vector<float>& state, size_t& sampledSteps)
{
  state.resize(mSize);
  kwCheck(kw_memcpy_d2h(ctx(), state.data(), mDeviceBuffer, mSize * sizeof(float)));
  sampledSteps = mCompressedTimeStep;
}
void IntensityAvgCOutputStream::restoreState(const float* state, size_t n, size_t sampledSteps)
{
  if (n != mSize) throw std::invalid_argument("checkpoint of stream " + mName + " has the wrong size");
  kwCheck(kw_memcpy_h2d(ctx(), mDeviceBuffer, state, mSize * sizeof(float)));
  mCompressedTimeStep = sampledSteps;
}
void IntensityAvgCOutputStream::create()
{
  mSize = mP.points();
  allocateMemory(); // zero-initialised device buffer
}
void IntensityAvgCOutputStream::postSample()
{
  const float* bufferP = mP.getCurrentStoreBuffer();
  const float* bufferU = mU.getCurrentStoreBuffer();
  if (bufferP && bufferU)
  {
    if (mP.is40bit())
      kwCheck(kw_intensity_avg_c_accumulate_40b(ctx(), mDeviceBuffer, bufferP, bufferU, mSize,
                                                static_cast<uint32_t>(CompressHelper::getInstance().getHarmonics()),
                                                mP.maxExp(), mU.maxExp()));
    else
    kwCheck(kw_intensity_avg_c_accumulate(ctx(), mDeviceBuffer, bufferP, bufferU, mSize,
                                          static_cast<uint32_t>(CompressHelper::getInstance().getHarmonics())));
    mCompressedTimeStep++;
  }
}
void IntensityAvgCOutputStream::accumulateStoredFrames(const float* frameP, const float* frameU)
{
  if (mP.is40bit()) // the reference's computeAverageIntensitiesC does not handle them either (:1541 "NOTE does not work ...")
    throw std::runtime_error("--post cannot compute I_avg_c / Q_term_c from 40-bit complex coefficients");
  const size_t floats = mSize * CompressHelper::getInstance().getHarmonics() * 2;
  void *dp = nullptr, *du = nullptr;
  kwCheck(kw_malloc(ctx(), floats * sizeof(float), &dp));
  kwCheck(kw_malloc(ctx(), floats * sizeof(float), &du));
  kwCheck(kw_memcpy_h2d(ctx(), dp, frameP, floats * sizeof(float)));
  kwCheck(kw_memcpy_h2d(ctx(), du, frameU, floats * sizeof(float)));
  kwCheck(kw_intensity_avg_c_accumulate(ctx(), mDeviceBuffer, static_cast<float*>(dp), static_cast<float*>(du), mSize,
                                        static_cast<uint32_t>(CompressHelper::getInstance().getHarmonics())));
  kwCheck(kw_sync(ctx()));
  kw_free(ctx(), dp);
  kw_free(ctx(), du);
  mCompressedTimeStep++;
}

void IntensityAvgCOutputStream::postProcess()
{ // IndexOutputStream.cpp:482-490
  if (mCompressedTimeStep > 0) kwCheck(kw_divide(ctx(), mDeviceBuffer, static_cast<float>(mCompressedTimeStep), mSize));
  mCompressedTimeStep = 0;
  copyAggregateFromDevice();
}

// ---- CuboidOutputStream ---------------------------------------------------------------------------------------------
void CuboidOutputStream::create()
{
  mSize = mSensorMask.getSizeOfAllCuboids();
  allocateMemory();
}
void CuboidOutputStream::sample()
{
  const DimensionSizes dims = mSourceMatrix.getDimensionSizes();
  size_t offset = 0;
  const int bb  = mSampledSteps & 1;
  float*    dst = (mReduceOp == ReduceOperator::kNone) ? mDeviceRaw[bb] : mDeviceBuffer;
  for (size_t c = 0; c < mSensorMask.getDimensionSizes().ny; c++)
  {
    const size_t n = mSensorMask.getSizeOfCuboid(c);
    OutputStreamsHipKernels::sampleCuboid(kernelOp(), dst + offset, mSourceMatrix.getDeviceData(),
                                          mSensorMask.getTopLeftCorner(c), mSensorMask.getBottomRightCorner(c), dims, n);
    offset += n;
  }
  if (mReduceOp == ReduceOperator::kNone) rawSampleTail(ctx(), dst, mPinned[bb], mEvent[bb], mSize);
  mSampledSteps++;
}
void CuboidOutputStream::flushRaw()
{
  if (mReduceOp != ReduceOperator::kNone || mFlushedSteps >= mSampledSteps) return;
  const int b = mFlushedSteps & 1;
  kwCheck(kw_event_synchronize(ctx(), mEvent[b]));
  storeRow(mPinned[b]);
  mFlushedSteps++;
}

// ---- WholeDomainOutputStream ----------------------------------------------------------------------------------------
void WholeDomainOutputStream::create()
{
  mSize = mSourceMatrix.size();
  allocateMemory();
}
void WholeDomainOutputStream::sample()
{
  OutputStreamsHipKernels::sampleAll(kernelOp(), mDeviceBuffer, mSourceMatrix.getDeviceData(), mSize);
  mSampledSteps++;
}

// ---- OutputStreamContainer ------------------------------------------------------------------------------------------
BaseOutputStream* OutputStreamContainer::createOutputStream(MatrixContainer& mc, MatrixContainer::MatrixIdx sampled,
                                                            const std::string& name, BaseOutputStream::ReduceOperator op)
{
  using MI = MatrixContainer::MatrixIdx;
  const Parameters& params = Parameters::getInstance();
  if (params.getSensorMaskType() == Parameters::SensorMaskType::kIndex)
    return new IndexOutputStream(name, mc.getMatrix<RealMatrix>(sampled), mc.getMatrix<IndexMatrix>(MI::kSensorMaskIndex), op);
  return new CuboidOutputStream(name, mc.getMatrix<RealMatrix>(sampled), mc.getMatrix<IndexMatrix>(MI::kSensorMaskCorners), op);
}

void OutputStreamContainer::init(MatrixContainer& mc)
{
  using OI = OutputStreamIdx;
  using MI = MatrixContainer::MatrixIdx;
  using RO = BaseOutputStream::ReduceOperator;
  const Parameters& params = Parameters::getInstance();
  const bool haveMask = mc.has(MI::kSensorMaskIndex) || mc.has(MI::kSensorMaskCorners);
  const bool is3D     = params.isSimulation3D(); // 2-D: no z-velocity streams (OutputStreamContainer.cpp:117-250)

  if (haveMask)
  {
    if (params.getStorePressureRawFlag()) mContainer[OI::kPressureRaw] = createOutputStream(mc, MI::kP, kPName, RO::kNone);
    if (params.getStorePressureRmsFlag()) mContainer[OI::kPressureRms] = createOutputStream(mc, MI::kP, kPRmsName, RO::kRms);
    if (params.getStorePressureMaxFlag()) mContainer[OI::kPressureMax] = createOutputStream(mc, MI::kP, kPMaxName, RO::kMax);
    if (params.getStorePressureMinFlag()) mContainer[OI::kPressureMin] = createOutputStream(mc, MI::kP, kPMinName, RO::kMin);
  }
  if (params.getStorePressureMaxAllFlag())
    mContainer[OI::kPressureMaxAll] = new WholeDomainOutputStream(kPMaxAllName, mc.getMatrix<RealMatrix>(MI::kP), RO::kMax);
  if (params.getStorePressureMinAllFlag())
    mContainer[OI::kPressureMinAll] = new WholeDomainOutputStream(kPMinAllName, mc.getMatrix<RealMatrix>(MI::kP), RO::kMin);
  if (haveMask)
  {
    if (params.getStoreVelocityRawFlag())
    {
      mContainer[OI::kVelocityXRaw] = createOutputStream(mc, MI::kUxSgx, kUxName, RO::kNone);
      mContainer[OI::kVelocityYRaw] = createOutputStream(mc, MI::kUySgy, kUyName, RO::kNone);
      if (is3D) mContainer[OI::kVelocityZRaw] = createOutputStream(mc, MI::kUzSgz, kUzName, RO::kNone);
    }
    if (params.getStoreVelocityNonStaggeredRawFlag())
    {
      mContainer[OI::kVelocityXNonStaggeredRaw] = createOutputStream(mc, MI::kUxShifted, kUxNonStaggeredName, RO::kNone);
      mContainer[OI::kVelocityYNonStaggeredRaw] = createOutputStream(mc, MI::kUyShifted, kUyNonStaggeredName, RO::kNone);
      if (is3D) mContainer[OI::kVelocityZNonStaggeredRaw] = createOutputStream(mc, MI::kUzShifted, kUzNonStaggeredName, RO::kNone);
    }
    struct Agg { bool on; RO op; const char* suffix; OI x, y, z; };
    const Agg aggs[] = {
      {params.getStoreVelocityRmsFlag(), RO::kRms, "_rms", OI::kVelocityXRms, OI::kVelocityYRms, OI::kVelocityZRms},
      {params.getStoreVelocityMaxFlag(), RO::kMax, "_max", OI::kVelocityXMax, OI::kVelocityYMax, OI::kVelocityZMax},
      {params.getStoreVelocityMinFlag(), RO::kMin, "_min", OI::kVelocityXMin, OI::kVelocityYMin, OI::kVelocityZMin}};
    for (const Agg& a : aggs)
      if (a.on)
      {
        mContainer[a.x] = createOutputStream(mc, MI::kUxSgx, kUxName + a.suffix, a.op);
        mContainer[a.y] = createOutputStream(mc, MI::kUySgy, kUyName + a.suffix, a.op);
        if (is3D) mContainer[a.z] = createOutputStream(mc, MI::kUzSgz, kUzName + a.suffix, a.op);
      }
  }
  // ---- average intensity / Q term from the stored raw series (OutputStreamContainer.cpp:227-271): the series they are
  // computed from are stored too, the intensities are kept out of the output unless asked for ----
  if (haveMask && (params.getStoreQTermFlag() || params.getStoreIntensityAvgFlag()))
  {
    if (mContainer.count(OI::kPressureRaw) == 0) mContainer[OI::kPressureRaw] = createOutputStream(mc, MI::kP, kPName, RO::kNone);
    if (mContainer.count(OI::kVelocityXNonStaggeredRaw) == 0)
    {
      mContainer[OI::kVelocityXNonStaggeredRaw] = createOutputStream(mc, MI::kUxShifted, kUxNonStaggeredName, RO::kNone);
      mContainer[OI::kVelocityYNonStaggeredRaw] = createOutputStream(mc, MI::kUyShifted, kUyNonStaggeredName, RO::kNone);
      if (is3D) mContainer[OI::kVelocityZNonStaggeredRaw] = createOutputStream(mc, MI::kUzShifted, kUzNonStaggeredName, RO::kNone);
    }
    const bool cuboid = !mc.has(MI::kSensorMaskIndex);
    const IndexMatrix& points = mc.getMatrix<IndexMatrix>(cuboid ? MI::kSensorMaskCorners : MI::kSensorMaskIndex);
    const bool hide = !params.getStoreIntensityAvgFlag();
    const RealMatrix& pm = mc.getMatrix<RealMatrix>(MI::kP);
    mContainer[OI::kIntensityXAvg] = new PostProcessedOutputStream("Ix_avg", pm, RO::kIAvg, points, cuboid, hide);
    mContainer[OI::kIntensityYAvg] = new PostProcessedOutputStream("Iy_avg", pm, RO::kIAvg, points, cuboid, hide);
    if (is3D) mContainer[OI::kIntensityZAvg] = new PostProcessedOutputStream("Iz_avg", pm, RO::kIAvg, points, cuboid, hide);
    if (params.getStoreQTermFlag()) mContainer[OI::kQTerm] = new PostProcessedOutputStream("Q_term", pm, RO::kQTerm, points, cuboid);
  }
  // ---- compression streams (OutputStreamContainer.cpp:92-96,157-168,272-321); index masks only ----
  const bool wantIAvgC = params.getStoreIntensityAvgCFlag() || params.getStoreQTermCFlag();
  if (mc.has(MI::kSensorMaskIndex) && params.getStoreVelocityCFlag())
  { // --u_c: the staggered velocities on the unshifted basis (OutputStreamContainer.cpp:133-142, BaseOutputStream.cpp:68-83)
    IndexMatrix& mask = mc.getMatrix<IndexMatrix>(MI::kSensorMaskIndex);
    mContainer[OI::kVelocityXC] = new CompressedIndexOutputStream(kUxName + "_c", mc.getMatrix<RealMatrix>(MI::kUxSgx), mask, false);
    mContainer[OI::kVelocityYC] = new CompressedIndexOutputStream(kUyName + "_c", mc.getMatrix<RealMatrix>(MI::kUySgy), mask, false);
    if (is3D) mContainer[OI::kVelocityZC] = new CompressedIndexOutputStream(kUzName + "_c", mc.getMatrix<RealMatrix>(MI::kUzSgz), mask, false);
  }
  if (mc.has(MI::kSensorMaskIndex) && (params.getStorePressureCFlag() || params.getStoreVelocityNonStaggeredCFlag() || wantIAvgC))
  {
    IndexMatrix& mask = mc.getMatrix<IndexMatrix>(MI::kSensorMaskIndex);
    if (params.getStorePressureCFlag() || wantIAvgC)
      mContainer[OI::kPressureC] = new CompressedIndexOutputStream(kPName + "_c", mc.getMatrix<RealMatrix>(MI::kP), mask, false);
    if (params.getStoreVelocityNonStaggeredCFlag() || wantIAvgC)
    {
      mContainer[OI::kVelocityXNonStaggeredC] = new CompressedIndexOutputStream(kUxNonStaggeredName + "_c", mc.getMatrix<RealMatrix>(MI::kUxShifted), mask, true);
      mContainer[OI::kVelocityYNonStaggeredC] = new CompressedIndexOutputStream(kUyNonStaggeredName + "_c", mc.getMatrix<RealMatrix>(MI::kUyShifted), mask, true);
      if (is3D) mContainer[OI::kVelocityZNonStaggeredC] = new CompressedIndexOutputStream(kUzNonStaggeredName + "_c", mc.getMatrix<RealMatrix>(MI::kUzShifted), mask, true);
    }
    if (wantIAvgC)
    {
      auto& pc = *static_cast<CompressedIndexOutputStream*>(mContainer[OI::kPressureC]);
      const OI us[3] = {OI::kVelocityXNonStaggeredC, OI::kVelocityYNonStaggeredC, OI::kVelocityZNonStaggeredC};
      const OI is[3] = {OI::kIntensityXAvgC, OI::kIntensityYAvgC, OI::kIntensityZAvgC};
      const char* names[3] = {"Ix_avg_c", "Iy_avg_c", "Iz_avg_c"};
      const int axes = is3D ? 3 : 2;
      for (int a = 0; a < axes; a++)
      {
        mContainer[is[a]] = new IntensityAvgCOutputStream(names[a], mc.getMatrix<RealMatrix>(MI::kP), pc,
                                                          *static_cast<CompressedIndexOutputStream*>(mContainer[us[a]]));
        mContainer[is[a]]->setDoNotSave(!params.getStoreIntensityAvgCFlag());
      }
      // coefficient series that only feed the intensities are not part of the output (:274-292: doNotSaveFlag)
      // (--post reads them from the output file: there they are ordinary stored streams, OutputStreamContainer.cpp:275-285)
      if (!params.getStorePressureCFlag() && !params.getOnlyPostProcessingFlag()) mContainer[OI::kPressureC]->setDoNotSave(true);
      if (!params.getStoreVelocityNonStaggeredCFlag() && !params.getOnlyPostProcessingFlag())
        for (int a = 0; a < axes; a++) mContainer[us[a]]->setDoNotSave(true);
      if (params.getStoreQTermCFlag())
        mContainer[OI::kQTermC] = new PostProcessedOutputStream("Q_term_c", mc.getMatrix<RealMatrix>(MI::kP), RO::kQTermC, mask, false);
    }
  }
  if (params.getStoreVelocityMaxAllFlag())
  {
    mContainer[OI::kVelocityXMaxAll] = new WholeDomainOutputStream(kUxName + "_max_all", mc.getMatrix<RealMatrix>(MI::kUxSgx), RO::kMax);
    mContainer[OI::kVelocityYMaxAll] = new WholeDomainOutputStream(kUyName + "_max_all", mc.getMatrix<RealMatrix>(MI::kUySgy), RO::kMax);
    if (is3D) mContainer[OI::kVelocityZMaxAll] = new WholeDomainOutputStream(kUzName + "_max_all", mc.getMatrix<RealMatrix>(MI::kUzSgz), RO::kMax);
  }
  if (params.getStoreVelocityMinAllFlag())
  {
    mContainer[OI::kVelocityXMinAll] = new WholeDomainOutputStream(kUxName + "_min_all", mc.getMatrix<RealMatrix>(MI::kUxSgx), RO::kMin);
    mContainer[OI::kVelocityYMinAll] = new WholeDomainOutputStream(kUyName + "_min_all", mc.getMatrix<RealMatrix>(MI::kUySgy), RO::kMin);
    if (is3D) mContainer[OI::kVelocityZMinAll] = new WholeDomainOutputStream(kUzName + "_min_all", mc.getMatrix<RealMatrix>(MI::kUzSgz), RO::kMin);
  }
}

void OutputStreamContainer::createStreams()
{
  for (auto& it : mContainer) it.second->create();
}
void OutputStreamContainer::sampleStreams()
{
  // index streams of the same field over the same mask (-p --p_max ...) share one launch; container order is kept
  // within a group, and every buffer receives exactly what its own sampleIndex<op> launch would write
  std::vector<BaseOutputStream*> done;
  for (auto& it : mContainer)
  {
    auto* first = dynamic_cast<IndexOutputStream*>(it.second);
    if (first == nullptr || std::find(done.begin(), done.end(), it.second) != done.end()) continue;
    std::vector<IndexOutputStream*> group{first};
    for (auto& jt : mContainer)
    {
      auto* other = dynamic_cast<IndexOutputStream*>(jt.second);
      if (other != nullptr && other != first && group.size() < 4 && &other->source() == &first->source() &&
          &other->mask() == &first->mask() && std::find(done.begin(), done.end(), jt.second) == done.end())
        group.push_back(other);
    }
    if (group.size() > 1)
    {
      OutputStreamsHipKernels::ReduceOperator ops[4];
      float* bufs[4];
      for (size_t g = 0; g < group.size(); g++) { ops[g] = group[g]->kernelOperator(); bufs[g] = group[g]->sampleTarget(); }
      OutputStreamsHipKernels::sampleIndexMulti(static_cast<int>(group.size()), ops, bufs, first->source().getDeviceData(),
                                                first->mask().getDeviceData(), first->size());
      for (IndexOutputStream* s : group) { s->sampleDone(); done.push_back(s); }
    }
  }
  for (auto& it : mContainer)
    if (std::find(done.begin(), done.end(), it.second) == done.end()) it.second->sample();
  // the reference runs these two passes inside the next step's flushRawStreams (OutputStreamContainer.cpp:380-403);
  // with device-resident accumulators they can follow the sampling immediately, in the same order
  for (auto& it : mContainer)
    if (auto* s = dynamic_cast<IntensityAvgCOutputStream*>(it.second)) s->postSample();
  for (auto& it : mContainer)
    if (auto* s = dynamic_cast<CompressedIndexOutputStream*>(it.second)) s->postSample2();
}
void OutputStreamContainer::flushRawStreams()
{
  for (auto& it : mContainer) it.second->flushRaw();
}
void OutputStreamContainer::postProcessStreams()
{
  for (auto& it : mContainer) it.second->postProcess();
}
void OutputStreamContainer::closeStreams()
{
  for (auto& it : mContainer) it.second->close();
}
void OutputStreamContainer::freeStreams()
{
  for (auto& it : mContainer) delete it.second;
  mContainer.clear();
}
BaseOutputStream* OutputStreamContainer::find(const std::string& name) const
{
  for (auto& it : mContainer)
    if (it.second->name() == name) return it.second;
  return nullptr;
}
std::vector<std::string> OutputStreamContainer::names(bool includeHidden) const
{
  std::vector<std::string> v;
  for (auto& it : mContainer)
    if (includeHidden || !it.second->doNotSave()) v.push_back(it.second->name());
  return v;
}
