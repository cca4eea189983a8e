// SeriesWriter.h — per-step output of sampled time series into the HDF5 output file.
//
// The reference's raw streams own a dataset of the output file and write one hyperslab per time step, one step late
// (IndexOutputStream.cpp:348-372 flushRaw -> flushBufferToFile, CuboidOutputStream.cpp:439-470,
// OutputStreamContainer.cpp:380-403); compression streams write one hyperslab per finished frame.  Here the output
// file is opened before the first step, every series stream gets a sink, and a writer thread appends the rows while the
// next steps are being enqueued: the host memory behind a series is a few queued rows instead of the whole
// (Nsens x Nt) array, and a run that dies keeps what it has sampled so far.
//
// HDF5 (as built in this image) is not thread-safe: while the writer thread exists every HDF5 call of the process goes
// through it or happens after drain() with the thread idle.
#ifndef KW_HOST_SERIES_WRITER_H
#define KW_HOST_SERIES_WRITER_H
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../OutputStreams.h"
#include "Hdf5File.h"

class Hdf5SeriesWriter
{
 public:
  /// one dataset a row is scattered to: floats [offset, offset + n) of the row go to dataset `name` at time index = row
  struct Part
  {
    std::string    name;
    DimensionSizes dims;   // (nx, ny, nz, nt): 4-D cuboid series; index series: (rowFloats, rows, 1) with nt == 0
    size_t         offset = 0, n = 0;
    hid_t          set = -1;
  };

  /// create == true: new output file (truncate), chunked like the reference's with deflate `compressionLevel`;
  /// create == false: re-open the output file of a checkpointed run and continue its datasets
  Hdf5SeriesWriter(const std::string& path, unsigned compressionLevel, bool create);
  ~Hdf5SeriesWriter();
  const std::string& path() const { return mPath; }
  unsigned compressionLevel() const { return mCompressionLevel; }
  Hdf5File& file() { return mFile; } // only after drain()

  /// sink for one stream; `parts` name the datasets (created now, or opened when the file was re-opened)
  std::unique_ptr<SeriesSink> makeSink(std::vector<Part> parts, size_t rowFloats, size_t totalRows);
  /// returns once every appended row is in the file and the writer thread is idle
  void drain();
  /// drain, stop the thread, close the datasets (the file stays open for the rest of the output)
  void finish();

 private:
  friend class Hdf5SeriesSink;
  struct Job
  {
    class Hdf5SeriesSink* sink;
    size_t                row;
    std::vector<float>    data;
  };
  void run();
  void submit(Job&& job);
  void writeRow(class Hdf5SeriesSink& sink, size_t row, const float* data);
  void readRows(class Hdf5SeriesSink& sink, size_t rows, std::vector<float>& out);

  std::string             mPath;
  unsigned                mCompressionLevel;
  bool                    mCreated;
  Hdf5File                mFile;
  std::thread             mThread;
  std::mutex              mMutex;
  std::condition_variable mWake, mIdle;
  std::deque<Job>         mQueue;
  size_t                  mQueuedFloats = 0;
  bool                    mBusy = false, mStop = false;
  std::string             mError; // first failure of the writer thread, rethrown on the caller's side
  std::vector<class Hdf5SeriesSink*> mSinks;
};
#endif
