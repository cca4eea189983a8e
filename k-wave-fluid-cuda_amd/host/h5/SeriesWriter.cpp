// SeriesWriter.cpp — see SeriesWriter.h.
#include "SeriesWriter.h"

#include <algorithm>
#include <ios>
#include <stdexcept>

namespace
{
constexpr size_t kChunkSize4MB   = 1048576;            // floats (IndexOutputStream.cpp:52, CuboidOutputStream.cpp:677)
constexpr size_t kMaxQueuedFloats = size_t(64) << 20;  // 256 MB of rows waiting for the disk, then append() blocks

void fail(const std::string& what) { throw std::ios_base::failure(what); }
} // namespace

/// One stream's rows -> its dataset(s).  Index / compression series: one 3-D dataset (1, rows, rowFloats), a row per
/// chunk (IndexOutputStream.cpp:119-126); cuboid series: one 4-D dataset (nt, nz, ny, nx) per cuboid, a time step per
/// chunk, cut into z-slabs above 32 MB (CuboidOutputStream.cpp:656-722).
class Hdf5SeriesSink : public SeriesSink
{
 public:
  Hdf5SeriesSink(Hdf5SeriesWriter& w, std::vector<Hdf5SeriesWriter::Part> parts, size_t rowFloats, size_t totalRows)
    : mWriter(w), mParts(std::move(parts)), mRowFloats(rowFloats), mTotalRows(totalRows) {}
  void append(const float* row, size_t floats) override
  {
    if (floats != mRowFloats) throw std::invalid_argument("series sink: row of the wrong length");
    if (mNextRow >= mTotalRows) throw std::runtime_error("series sink: more rows than the dataset " + mParts[0].name + " holds");
    Hdf5SeriesWriter::Job job{this, mNextRow++, std::vector<float>(row, row + floats)};
    mWriter.submit(std::move(job));
  }
  void flush() override { mWriter.drain(); }
  void read(std::vector<float>& out, size_t rows) override
  {
    mWriter.drain();
    mWriter.readRows(*this, rows, out);
  }
  void setRows(size_t rows) override
  {
    if (rows > mTotalRows) throw std::invalid_argument("series sink: the output file is shorter than the checkpoint says");
    mNextRow = rows;
  }
  Hdf5SeriesWriter&                   mWriter;
  std::vector<Hdf5SeriesWriter::Part> mParts;
  size_t                              mRowFloats, mTotalRows, mNextRow = 0;
};

Hdf5SeriesWriter::Hdf5SeriesWriter(const std::string& path, unsigned compressionLevel, bool create)
  : mPath(path), mCompressionLevel(compressionLevel), mCreated(create)
{
  if (create) mFile.create(path);
  else mFile.open(path, false);
  mFile.setOutputLayout(true, compressionLevel);
  mThread = std::thread([this] { run(); });
}

Hdf5SeriesWriter::~Hdf5SeriesWriter()
{
  try { finish(); } catch (...) {}
}

std::unique_ptr<SeriesSink> Hdf5SeriesWriter::makeSink(std::vector<Part> parts, size_t rowFloats, size_t totalRows)
{
  drain(); // the thread is idle: HDF5 calls from this thread are safe
  for (Part& p : parts)
  {
    const bool     cuboid = p.dims.nt > 0;
    const int      rank   = cuboid ? 4 : 3;
    hsize_t dims[4], chunk[4];
    if (cuboid)
    {
      dims[0] = p.dims.nt; dims[1] = p.dims.nz; dims[2] = p.dims.ny; dims[3] = p.dims.nx;
      chunk[0] = 1; chunk[1] = p.dims.nz; chunk[2] = p.dims.ny; chunk[3] = p.dims.nx;
      if (p.dims.nx * p.dims.ny * p.dims.nz > kChunkSize4MB * 8)
      {
        size_t nSlabs = 1;
        while (nSlabs * p.dims.nx * p.dims.ny < kChunkSize4MB) nSlabs++;
        chunk[1] = nSlabs;
      }
    }
    else
    {
      dims[0] = 1; dims[1] = totalRows; dims[2] = rowFloats;
      chunk[0] = 1; chunk[1] = 1; chunk[2] = (rowFloats > kChunkSize4MB * 8) ? kChunkSize4MB : rowFloats;
    }
    if (mFile.datasetExists(p.name))
    {
      p.set = H5Dopen2(mFile.handle(), p.name.c_str(), H5P_DEFAULT);
      if (p.set < 0) fail("Error: cannot open dataset \"" + p.name + "\" of the output file");
      hid_t sp = H5Dget_space(p.set);
      hsize_t have[4] = {0, 0, 0, 0};
      const bool same = (H5Sget_simple_extent_ndims(sp) == rank) && (H5Sget_simple_extent_dims(sp, have, nullptr) == rank) &&
                        std::equal(have, have + rank, dims);
      H5Sclose(sp);
      if (!same) fail("Error: dataset \"" + p.name + "\" of the output file does not match this simulation");
    }
    else
    {
      if (!mCreated) fail("Error: the output file to continue holds no dataset \"" + p.name + "\"");
      hid_t space = H5Screate_simple(rank, dims, nullptr);
      hid_t plist = H5Pcreate(H5P_DATASET_CREATE);
      if (rowFloats > 0 && totalRows > 0)
      {
        H5Pset_chunk(plist, rank, chunk);
        if (mCompressionLevel > 0) H5Pset_deflate(plist, mCompressionLevel);
      }
      p.set = H5Dcreate2(mFile.handle(), p.name.c_str(), H5T_IEEE_F32LE, space, H5P_DEFAULT, plist, H5P_DEFAULT);
      H5Pclose(plist);
      H5Sclose(space);
      if (p.set < 0) fail("Error: cannot create dataset \"" + p.name + "\"");
      mFile.writeStringAttribute(p.name, Hdf5File::kMatrixDataTypeName, "float");
      mFile.writeStringAttribute(p.name, Hdf5File::kMatrixDomainTypeName, "real");
    }
  }
  std::unique_ptr<Hdf5SeriesSink> sink(new Hdf5SeriesSink(*this, std::move(parts), rowFloats, totalRows));
  mSinks.push_back(sink.get());
  return sink;
}

void Hdf5SeriesWriter::writeRow(Hdf5SeriesSink& sink, size_t row, const float* data)
{
  for (const Part& p : sink.mParts)
  {
    const bool cuboid = p.dims.nt > 0;
    const int  rank   = cuboid ? 4 : 3;
    hsize_t start[4] = {0, 0, 0, 0}, count[4];
    if (cuboid) { start[0] = row; count[0] = 1; count[1] = p.dims.nz; count[2] = p.dims.ny; count[3] = p.dims.nx; }
    else { start[1] = row; count[0] = 1; count[1] = 1; count[2] = sink.mRowFloats; }
    hid_t fspace = H5Dget_space(p.set);
    H5Sselect_hyperslab(fspace, H5S_SELECT_SET, start, nullptr, count, nullptr);
    hid_t mspace = H5Screate_simple(rank, count, nullptr);
    const herr_t st = H5Dwrite(p.set, H5T_NATIVE_FLOAT, mspace, fspace, H5P_DEFAULT, data + p.offset);
    H5Sclose(mspace);
    H5Sclose(fspace);
    if (st < 0) fail("Error: cannot write time step " + std::to_string(row) + " of dataset \"" + p.name + "\"");
  }
}

void Hdf5SeriesWriter::readRows(Hdf5SeriesSink& sink, size_t rows, std::vector<float>& out)
{ // caller has drained: the thread is idle
  out.assign(rows * sink.mRowFloats, 0.0f);
  if (rows == 0) return;
  std::vector<float> block;
  for (const Part& p : sink.mParts)
  {
    const bool cuboid = p.dims.nt > 0;
    const int  rank   = cuboid ? 4 : 3;
    hsize_t start[4] = {0, 0, 0, 0}, count[4];
    if (cuboid) { count[0] = rows; count[1] = p.dims.nz; count[2] = p.dims.ny; count[3] = p.dims.nx; }
    else { count[0] = 1; count[1] = rows; count[2] = sink.mRowFloats; }
    hid_t fspace = H5Dget_space(p.set);
    H5Sselect_hyperslab(fspace, H5S_SELECT_SET, start, nullptr, count, nullptr);
    hid_t mspace = H5Screate_simple(rank, count, nullptr);
    herr_t st;
    if (!cuboid) st = H5Dread(p.set, H5T_NATIVE_FLOAT, mspace, fspace, H5P_DEFAULT, out.data());
    else
    { // the row of a cuboid stream holds the cuboids back to back
      block.resize(rows * p.n);
      st = H5Dread(p.set, H5T_NATIVE_FLOAT, mspace, fspace, H5P_DEFAULT, block.data());
      for (size_t t = 0; t < rows; t++) std::copy_n(block.data() + t * p.n, p.n, out.data() + t * sink.mRowFloats + p.offset);
    }
    H5Sclose(mspace);
    H5Sclose(fspace);
    if (st < 0) fail("Error: cannot read dataset \"" + p.name + "\" back from the output file");
  }
}

void Hdf5SeriesWriter::submit(Job&& job)
{
  std::unique_lock<std::mutex> lock(mMutex);
  if (!mError.empty()) fail(mError);
  mIdle.wait(lock, [&] { return mQueuedFloats < kMaxQueuedFloats || !mError.empty(); });
  mQueuedFloats += job.data.size();
  mQueue.push_back(std::move(job));
  mWake.notify_one();
}

void Hdf5SeriesWriter::run()
{
  std::unique_lock<std::mutex> lock(mMutex);
  for (;;)
  {
    mWake.wait(lock, [&] { return mStop || !mQueue.empty(); });
    if (mQueue.empty()) { if (mStop) return; continue; }
    Job job = std::move(mQueue.front());
    mQueue.pop_front();
    mBusy = true;
    lock.unlock();
    std::string err;
    try { writeRow(*job.sink, job.row, job.data.data()); }
    catch (const std::exception& e) { err = e.what(); }
    lock.lock();
    mQueuedFloats -= job.data.size();
    mBusy = false;
    if (!err.empty() && mError.empty()) mError = err;
    mIdle.notify_all();
  }
}

void Hdf5SeriesWriter::drain()
{
  std::unique_lock<std::mutex> lock(mMutex);
  mIdle.wait(lock, [&] { return mQueue.empty() && !mBusy; });
  if (!mError.empty()) fail(mError);
}

void Hdf5SeriesWriter::finish()
{
  if (mThread.joinable())
  {
    {
      std::unique_lock<std::mutex> lock(mMutex);
      mIdle.wait(lock, [&] { return mQueue.empty() && !mBusy; });
      mStop = true;
      mWake.notify_all();
    }
    mThread.join();
  }
  for (Hdf5SeriesSink* s : mSinks)
    for (Part& p : s->mParts)
    {
      if (p.set >= 0) H5Dclose(p.set);
      p.set = -1;
    }
  mSinks.clear();
  if (!mError.empty()) fail(mError);
}
