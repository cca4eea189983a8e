// OutputStreams.h — sampling streams on the caller side of the sampling kernels.
// Mirror of OutputStreams/{Base,Index,Cuboid,WholeDomain}OutputStream.{h,cpp} and
// Containers/OutputStreamContainer.{h,cpp}, restricted to what sits on the per-step path:
// sample() -> kernel call sites, reduce-operator initial values (BaseOutputStream.cpp:271-367), the
// one-step-delayed flush of raw series (KSpaceFirstOrderSolver.cpp:1060-1093) and RMS post-processing.
// Raw series land in pinned double buffers (async D2H + event) instead of the reference's zero-copy mapped
// buffer (BaseOutputStream.cpp:369-388); the HDF5 dataset behind a stream is replaced by an in-memory dataset
// (std::vector) that the HDF5 writer (optional component) or Python reads out.
#ifndef KW_HOST_OUTPUT_STREAMS_H
#define KW_HOST_OUTPUT_STREAMS_H
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "MatrixContainer.h"

/// Replaces namespace OutputStreamsCudaKernels (OutputStreams/OutputStreamsCudaKernels.cuh:47-106)
namespace OutputStreamsHipKernels
{
enum class ReduceOperator { kNone = 0, kRms = 1, kMax = 2, kMin = 3 };
void sampleIndex(ReduceOperator op, float* samplingBuffer, const float* sourceData, const size_t* sensorData, size_t nSamples);
void sampleIndexMulti(int nOps, const ReduceOperator* ops, float* const* samplingBuffers, const float* sourceData,
                      const size_t* sensorData, size_t nSamples);
void sampleCuboid(ReduceOperator op, float* samplingBuffer, const float* sourceData, const DimensionSizes& topLeftCorner,
                  const DimensionSizes& bottomRightCorner, const DimensionSizes& matrixSize, size_t nSamples);
void sampleAll(ReduceOperator op, float* samplingBuffer, const float* sourceData, size_t nSamples);
void postProcessingRms(float* samplingBuffer, float scalingCoeff, size_t nSamples);
} // namespace OutputStreamsHipKernels

/// Where the rows of a stored time series go.  Without a sink a series stream keeps its rows in memory (dataset());
/// the HDF5 component installs sinks that append every row to the output file as it is flushed (h5/SeriesWriter.h),
/// which is what the reference's streams do with their own dataset (IndexOutputStream.cpp:348-372).
class SeriesSink
{
 public:
  virtual ~SeriesSink() = default;
  virtual void append(const float* row, size_t floats) = 0;       ///< next row; may return before it is on disk
  virtual void flush() = 0;                                        ///< every appended row is in the file on return
  virtual void read(std::vector<float>& out, size_t rows) = 0;    ///< rows [0, rows) back from the file
  virtual void setRows(size_t rows) = 0;                           ///< restart: `rows` rows are in the file already
};

class BaseOutputStream
{
 public:
  enum class ReduceOperator { kNone, kRms, kMax, kMin, kC, kIAvgC, kIAvg, kQTerm, kQTermC };
  BaseOutputStream(const std::string& name, const RealMatrix& source, ReduceOperator op, bool doNotSave = false)
    : mName(name), mSourceMatrix(source), mReduceOp(op), mDoNotSave(doNotSave) {}
  virtual ~BaseOutputStream();
  virtual void create() = 0;
  virtual void sample() = 0;
  virtual void flushRaw() {}
  virtual void postProcess();
  virtual void close() {}
  const std::string& name() const { return mName; }
  ReduceOperator     reduceOp() const { return mReduceOp; }
  /// stored dataset: raw = [sampledSteps][mSize]; aggregated = [mSize] (valid after postProcess).  A series that goes
  /// to a sink is not held here: loadSeries() reads it back for the few users that need all of it (I_avg, stream_read)
  const std::vector<float>& dataset() const { return mDataset; }
  bool isSeries() const { return mReduceOp == ReduceOperator::kNone || mReduceOp == ReduceOperator::kC; }
  void attachSink(std::unique_ptr<SeriesSink> sink) { mSink = std::move(sink); std::vector<float>().swap(mDataset); }
  bool hasSink() const { return static_cast<bool>(mSink); }
  void loadSeries();
  void adoptStoredSeries(size_t rows); ///< --post: `rows` rows of this series are in the (re-opened) output file
  void releaseSeries() { if (mSink) std::vector<float>().swap(mDataset); }
  size_t size() const { return mSize; }
  size_t sampledSteps() const { return mFlushedSteps; }
  /// stream that only feeds another one (the reference's doNotSaveFlag, e.g. I_avg behind --Q_term): not listed, not written
  bool   doNotSave() const { return mDoNotSave; }
  void   setDoNotSave(bool v) { mDoNotSave = v; }
  /// Checkpoint / restart (BaseOutputStream::checkpoint / reopen, e.g. IndexOutputStream.cpp:497-533): what a restart
  /// needs to continue this stream — the series stored so far (raw; the reference keeps it in the output file) or the
  /// device accumulator (rms / max / min) — and the number of sampled steps.
  virtual void checkpointState(std::vector<float>& state, size_t& sampledSteps);
  virtual void restoreState(const float* state, size_t n, size_t sampledSteps);

 protected:
  void allocateMemory();
  void freeMemory();
  void storeRow(const float* row); ///< one flushed row of a series: to the sink, else appended to the in-memory dataset
  void copyAggregateFromDevice();
  OutputStreamsHipKernels::ReduceOperator kernelOp() const;

  std::string       mName;
  const RealMatrix& mSourceMatrix;
  ReduceOperator    mReduceOp;
  bool              mDoNotSave = false;
  size_t            mSize = 0;
  float*            mDeviceBuffer = nullptr;       // aggregate, or raw staging buffer 0 (device)
  float*            mDeviceRaw[2] = {nullptr, nullptr}; // raw: double-buffered device staging (copy overlaps compute)
  float*            mPinned[2]    = {nullptr, nullptr}; // raw: pinned double buffer
  void*             mEvent[2]     = {nullptr, nullptr};
  size_t            mSampledSteps = 0, mFlushedSteps = 0;
  std::vector<float> mDataset;
  std::unique_ptr<SeriesSink> mSink;
};

class IndexOutputStream : public BaseOutputStream
{
 public:
  IndexOutputStream(const std::string& name, const RealMatrix& source, const IndexMatrix& sensorMask, ReduceOperator op)
    : BaseOutputStream(name, source, op), mSensorMask(sensorMask) {}
  void create() override;
  void sample() override;   // IndexOutputStream.cpp:253-293
  void flushRaw() override; // IndexOutputStream.cpp:348-371 (raw branch)
  /// sample() in two halves, so that the container can serve several streams of one field with one kernel launch:
  /// where this step's values go and with which operator / what follows the kernel (raw: the D2H copy)
  float* sampleTarget();
  void   sampleDone();
  const RealMatrix&  source() const { return mSourceMatrix; }
  const IndexMatrix& mask() const { return mSensorMask; }
  OutputStreamsHipKernels::ReduceOperator kernelOperator() const { return kernelOp(); }
 private:
  const IndexMatrix& mSensorMask;
};

/// kC streams: on-the-fly compression of the sampled series (IndexOutputStream.cpp:373-470), accumulators on the device.
class CompressedIndexOutputStream : public BaseOutputStream
{
 public:
  CompressedIndexOutputStream(const std::string& name, const RealMatrix& source, const IndexMatrix& sensorMask,
                              bool shiftedBasis)
    : BaseOutputStream(name, source, ReduceOperator::kC), mSensorMask(sensorMask), mShifted(shiftedBasis) {}
  ~CompressedIndexOutputStream() override;
  void create() override;
  void sample() override;       // gather + correlate (flushRaw's kC branch, done at sampling time on the device)
  void postSample2();           // emit the finished frame, zero its accumulator (BaseOutputStream.cpp:117-132)
  /// device buffer holding the frame finished at this sampled step, or nullptr (getCurrentStoreBuffer)
  const float* getCurrentStoreBuffer() const { return mCurrent; }
  /// state = frames stored so far, then the accumulators c1 (and c2 unless --no_overlap); steps = sampled steps
  void checkpointState(std::vector<float>& state, size_t& sampledSteps) override;
  void restoreState(const float* state, size_t n, size_t sampledSteps) override;
  /// the reference's checkpoint form (BaseOutputStream.cpp:551-606): the two accumulators alone — Temp_<name>_1 / _2; with
  /// --no_overlap they are one buffer and c2 == c1.  restoreAccumulators: the frames so far are in the re-opened output
  /// file (or, for a stream that only feeds others, nowhere: nothing reads them again)
  void accumulators(std::vector<float>& c1, std::vector<float>& c2);
  void restoreAccumulators(const float* c1, const float* c2, size_t n, size_t sampledSteps);
  size_t frames() const { return mCompressedTimeStep; }
  size_t points() const { return mSensorMask.size(); }
  bool   shiftedBasis() const { return mShifted; }
  bool   is40bit() const { return m40bit; }
  int    maxExp() const; ///< exponent bias of the 40-bit format: 138, or 114 on the shifted (velocity) basis

 private:
  const IndexMatrix& mSensorMask;
  bool    mShifted;
  float*  mC1 = nullptr;
  float*  mC2 = nullptr;
  float*  mBE = nullptr;
  float*  mBE_1 = nullptr;
  float*  mCurrent = nullptr;
  bool    mSavingFlag = false;
  bool    m40bit = false; // --40-bit_complex: accumulators and frames are 5-byte packed complex numbers
  size_t  mCompressedTimeStep = 0;
  std::vector<float> mFrameHost;
};

/// kIAvgC streams: time-averaged intensity from the compression coefficients (IndexOutputStream.cpp:299-342,482-490)
class IntensityAvgCOutputStream : public BaseOutputStream
{
 public:
  IntensityAvgCOutputStream(const std::string& name, const RealMatrix& source, const CompressedIndexOutputStream& p,
                            const CompressedIndexOutputStream& u)
    : BaseOutputStream(name, source, ReduceOperator::kIAvgC), mP(p), mU(u) {}
  void create() override;
  void sample() override {}
  void postSample();
  void postProcess() override;
  /// --post: one stored pair of coefficient frames (host) added like postSample() adds the device ones
  void accumulateStoredFrames(const float* frameP, const float* frameU);
  /// state = the running sum; steps = number of frames added so far
  void checkpointState(std::vector<float>& state, size_t& sampledSteps) override;
  void restoreState(const float* state, size_t n, size_t sampledSteps) override;

 private:
  const CompressedIndexOutputStream& mP;
  const CompressedIndexOutputStream& mU;
  size_t mCompressedTimeStep = 0;
};

/// kIAvg / kQTerm / kQTermC streams: one value per sensor point, produced after the last step by the solver's
/// post-processing (computeAverageIntensities / computeQTerm) instead of by sampling
class PostProcessedOutputStream : public BaseOutputStream
{
 public:
  PostProcessedOutputStream(const std::string& name, const RealMatrix& source, ReduceOperator op,
                            const IndexMatrix& sensorMask, bool cuboidMask, bool doNotSave = false)
    : BaseOutputStream(name, source, op, doNotSave), mSensorMask(sensorMask), mCuboidMask(cuboidMask) {}
  void create() override
  { // the mask is loaded by now (the container is set up before the input file is read)
    mSize = mCuboidMask ? mSensorMask.getSizeOfAllCuboids() : mSensorMask.size();
    mDataset.assign(mSize, 0.0f);
  }
  void sample() override {}
  void postProcess() override {}
  std::vector<float>& data() { return mDataset; }
  void checkpointState(std::vector<float>& state, size_t& sampledSteps) override { state.clear(); sampledSteps = 0; }
  void restoreState(const float*, size_t, size_t) override {}

 private:
  const IndexMatrix& mSensorMask;
  bool               mCuboidMask;
};

class CuboidOutputStream : public BaseOutputStream
{
 public:
  CuboidOutputStream(const std::string& name, const RealMatrix& source, const IndexMatrix& sensorMask, ReduceOperator op)
    : BaseOutputStream(name, source, op), mSensorMask(sensorMask) {}
  void create() override;
  void sample() override;   // CuboidOutputStream.cpp:263-345: one launch per cuboid
  void flushRaw() override;
 private:
  const IndexMatrix& mSensorMask;
};

class WholeDomainOutputStream : public BaseOutputStream
{
 public:
  WholeDomainOutputStream(const std::string& name, const RealMatrix& source, ReduceOperator op)
    : BaseOutputStream(name, source, op) {}
  void create() override;
  void sample() override; // WholeDomainOutputStream.cpp:143-198
};

class OutputStreamContainer
{
 public:
  /// order is semantic (OutputStreamContainer.h:56-57): iteration follows the enum
  enum class OutputStreamIdx
  {
    kPressureRaw, kPressureRms, kPressureMax, kPressureMin, kPressureMaxAll, kPressureMinAll,
    kVelocityXRaw, kVelocityYRaw, kVelocityZRaw, kVelocityXNonStaggeredRaw, kVelocityYNonStaggeredRaw,
    kVelocityZNonStaggeredRaw, kVelocityXRms, kVelocityYRms, kVelocityZRms, kVelocityXMax, kVelocityYMax,
    kVelocityZMax, kVelocityXMin, kVelocityYMin, kVelocityZMin, kVelocityXMaxAll, kVelocityYMaxAll, kVelocityZMaxAll,
    kVelocityXMinAll, kVelocityYMinAll, kVelocityZMinAll,
    kPressureC, kVelocityXNonStaggeredC, kVelocityYNonStaggeredC, kVelocityZNonStaggeredC,
    kIntensityXAvgC, kIntensityYAvgC, kIntensityZAvgC,
    kIntensityXAvg, kIntensityYAvg, kIntensityZAvg, kQTerm, kQTermC,
    kVelocityXC, kVelocityYC, kVelocityZC
  };
  ~OutputStreamContainer() { freeStreams(); }
  void init(MatrixContainer& matrixContainer); // OutputStreamContainer.cpp:70-325
  void createStreams();
  void sampleStreams();      // :364-373
  void flushRawStreams();    // :380-403
  void postProcessStreams(); // :339-345
  void closeStreams();
  void freeStreams();
  bool empty() const { return mContainer.empty(); }
  BaseOutputStream* find(const std::string& name) const;
  BaseOutputStream* get(OutputStreamIdx idx) const { auto it = mContainer.find(idx); return it == mContainer.end() ? nullptr : it->second; }
  /// streams that are part of the output; includeHidden adds the ones that only feed others (checkpointing needs all)
  std::vector<std::string> names(bool includeHidden = false) const;

 private:
  BaseOutputStream* createOutputStream(MatrixContainer& mc, MatrixContainer::MatrixIdx sampled, const std::string& name,
                                       BaseOutputStream::ReduceOperator op);
  std::map<OutputStreamIdx, BaseOutputStream*> mContainer;
};
#endif
