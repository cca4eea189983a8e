// Hdf5File.cpp — see Hdf5File.h.
#include "Hdf5File.h"

#include <ctime>
#include <ios>
#include <stdexcept>

const std::string Hdf5File::kMatrixDomainTypeName = "domain_type";
const std::string Hdf5File::kMatrixDataTypeName   = "data_type";
static const char* kDomainNames[] = {"real", "complex"};
static const char* kDataNames[]   = {"float", "long"};

static void fail(const std::string& what) { throw std::ios_base::failure(what); }

Hdf5File::~Hdf5File() { close(); }

void Hdf5File::create(const std::string& fileName)
{
  close();
  mFile = H5Fcreate(fileName.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  if (mFile < 0) fail("Error: File \"" + fileName + "\" could not be created");
  mName = fileName;
}
void Hdf5File::open(const std::string& fileName, bool readOnly)
{
  close();
  if (H5Fis_hdf5(fileName.c_str()) <= 0) fail("Error: File \"" + fileName + "\" is not a valid HDF5 file or does not exist");
  mFile = H5Fopen(fileName.c_str(), readOnly ? H5F_ACC_RDONLY : H5F_ACC_RDWR, H5P_DEFAULT);
  if (mFile < 0) fail("Error: File \"" + fileName + "\" could not be opened");
  mName = fileName;
}
void Hdf5File::close()
{
  if (mFile >= 0) H5Fclose(mFile);
  mFile = -1;
}
bool Hdf5File::datasetExists(const std::string& name) const { return H5Lexists(mFile, name.c_str(), H5P_DEFAULT) > 0; }

DimensionSizes Hdf5File::getDatasetDimensionSizes(const std::string& name) const
{
  int rank = 0;
  if (H5LTget_dataset_ndims(mFile, name.c_str(), &rank) < 0) fail("Error: cannot read dimension sizes of dataset \"" + name + "\"");
  std::vector<hsize_t> dims(rank, 1);
  if (H5LTget_dataset_info(mFile, name.c_str(), dims.data(), nullptr, nullptr) < 0)
    fail("Error: cannot read dimension sizes of dataset \"" + name + "\"");
  if (rank == 3) return DimensionSizes(dims[2], dims[1], dims[0]);
  if (rank == 4 && dims[0] == 0) fail("Error: dataset \"" + name + "\" is empty");
  if (rank == 4) return DimensionSizes(dims[3], dims[2], dims[1], dims[0]);
  fail("Error: dataset \"" + name + "\" is not 3-D / 4-D");
  return DimensionSizes();
}
size_t Hdf5File::getDatasetSize(const std::string& name) const { return getDatasetDimensionSizes(name).nElements(); }

std::string Hdf5File::readStringAttribute(const std::string& dataset, const std::string& attr) const
{
  char buf[256] = {0};
  if (H5LTget_attribute_string(mFile, dataset.c_str(), attr.c_str(), buf) < 0)
    fail("Error: cannot read attribute \"" + attr + "\" of \"" + dataset + "\"");
  return std::string(buf);
}
void Hdf5File::writeStringAttribute(const std::string& dataset, const std::string& attr, const std::string& value)
{
  if (H5LTset_attribute_string(mFile, dataset.c_str(), attr.c_str(), value.c_str()) < 0)
    fail("Error: cannot write attribute \"" + attr + "\" of \"" + dataset + "\"");
}
void Hdf5File::writeLongLongAttribute(const std::string& dataset, const std::string& attr, long long value)
{
  if (H5LTset_attribute_long_long(mFile, dataset.c_str(), attr.c_str(), &value, 1) < 0)
    fail("Error: cannot write attribute \"" + attr + "\" of \"" + dataset + "\"");
}
void Hdf5File::writeFloatAttribute(const std::string& dataset, const std::string& attr, float value)
{
  if (H5LTset_attribute_float(mFile, dataset.c_str(), attr.c_str(), &value, 1) < 0)
    fail("Error: cannot write attribute \"" + attr + "\" of \"" + dataset + "\"");
}
double Hdf5File::readNumericAttribute(const std::string& dataset, const std::string& attr) const
{
  double v = 0.0;
  if (H5LTget_attribute_double(mFile, dataset.c_str(), attr.c_str(), &v) < 0)
    fail("Error: cannot read attribute \"" + attr + "\" of \"" + dataset + "\"");
  return v;
}
Hdf5File::MatrixDataType Hdf5File::readMatrixDataType(const std::string& name) const
{
  const std::string v = readStringAttribute(name, kMatrixDataTypeName);
  if (v == kDataNames[0]) return MatrixDataType::kFloat;
  if (v == kDataNames[1]) return MatrixDataType::kLong;
  fail("Error: bad data_type attribute of dataset \"" + name + "\"");
  return MatrixDataType::kFloat;
}
Hdf5File::MatrixDomainType Hdf5File::readMatrixDomainType(const std::string& name) const
{
  const std::string v = readStringAttribute(name, kMatrixDomainTypeName);
  if (v == kDomainNames[0]) return MatrixDomainType::kReal;
  if (v == kDomainNames[1]) return MatrixDomainType::kComplex;
  fail("Error: bad domain_type attribute of dataset \"" + name + "\"");
  return MatrixDomainType::kReal;
}
void Hdf5File::readCompleteDataset(const std::string& name, size_t nElements, float* data) const
{
  if (getDatasetSize(name) != nElements) fail("Error: dataset \"" + name + "\" has wrong dimension sizes");
  if (nElements == 0) return;
  if (H5LTread_dataset(mFile, name.c_str(), H5T_NATIVE_FLOAT, data) < 0) fail("Error: cannot read dataset \"" + name + "\"");
}
void Hdf5File::readCompleteDataset(const std::string& name, size_t nElements, size_t* data) const
{
  if (getDatasetSize(name) != nElements) fail("Error: dataset \"" + name + "\" has wrong dimension sizes");
  if (nElements == 0) return;
  if (H5LTread_dataset(mFile, name.c_str(), H5T_NATIVE_UINT64, data) < 0) fail("Error: cannot read dataset \"" + name + "\"");
}
void Hdf5File::readPlanes(const std::string& name, size_t z0, size_t nPlanes, float* data) const
{
  const DimensionSizes d = getDatasetDimensionSizes(name);
  if (d.nt > 0 || z0 + nPlanes > d.nz) fail("Error: dataset \"" + name + "\" does not hold the requested planes");
  if (nPlanes == 0) return;
  hid_t set = H5Dopen2(mFile, name.c_str(), H5P_DEFAULT);
  if (set < 0) fail("Error: cannot open dataset \"" + name + "\"");
  hid_t fspace = H5Dget_space(set);
  const hsize_t start[3] = {z0, 0, 0}, count[3] = {nPlanes, d.ny, d.nx};
  H5Sselect_hyperslab(fspace, H5S_SELECT_SET, start, nullptr, count, nullptr);
  hid_t mspace = H5Screate_simple(3, count, nullptr);
  const herr_t st = H5Dread(set, H5T_NATIVE_FLOAT, mspace, fspace, H5P_DEFAULT, data);
  H5Sclose(mspace);
  H5Sclose(fspace);
  H5Dclose(set);
  if (st < 0) fail("Error: cannot read dataset \"" + name + "\"");
}
// Chunking of output datasets as the reference lays them out: one z-plane per chunk for 2-D / 3-D data, 256 KiB - 4 MiB
// pieces for long 1-D data (RealMatrix.cpp:88-110), deflate at the requested level (Hdf5File.cpp:330-350).
static DimensionSizes chunkSizes(const DimensionSizes& d)
{
  DimensionSizes c = d;
  c.nz = 1;
  if (d.ny == 1 && d.nz == 1)
  {
    constexpr size_t k4MB = 1048576, k1MB = 262144, k256kB = 65536;
    if (d.nx > 4 * k4MB) c.nx = k4MB;
    else if (d.nx > 4 * k1MB) c.nx = k1MB;
    else if (d.nx > 4 * k256kB) c.nx = k256kB;
  }
  return c;
}

static void writeDataset(hid_t file, const std::string& name, const DimensionSizes& d, hid_t fileType, hid_t memType,
                         const void* data, bool chunked = false, unsigned compressionLevel = 0)
{
  hsize_t dims[3] = {d.nz, d.ny, d.nx};
  if (H5Lexists(file, name.c_str(), H5P_DEFAULT) > 0)
  { // rewritten in place (the accumulators a checkpoint leaves in the output file, the scalars of every leg): same extent
    hid_t set = H5Dopen2(file, name.c_str(), H5P_DEFAULT);
    herr_t st = -1;
    if (set >= 0)
    {
      hid_t sp = H5Dget_space(set);
      hsize_t have[3] = {0, 0, 0};
      const bool same = (H5Sget_simple_extent_ndims(sp) == 3) && (H5Sget_simple_extent_dims(sp, have, nullptr) == 3) &&
                        have[0] == dims[0] && have[1] == dims[1] && have[2] == dims[2];
      H5Sclose(sp);
      if (same) st = (d.nx * d.ny * d.nz == 0) ? 0 : H5Dwrite(set, memType, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
      H5Dclose(set);
    }
    if (st < 0) throw std::ios_base::failure("Error: cannot rewrite dataset \"" + name + "\"");
    return;
  }
  hid_t space = H5Screate_simple(3, dims, nullptr);
  hid_t plist = H5P_DEFAULT;
  if (chunked && d.nElements() > 0)
  {
    const DimensionSizes c = chunkSizes(d);
    hsize_t cdims[3] = {c.nz, c.ny, c.nx};
    plist = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_chunk(plist, 3, cdims);
    if (compressionLevel > 0) H5Pset_deflate(plist, compressionLevel);
  }
  hid_t set   = H5Dcreate2(file, name.c_str(), fileType, space, H5P_DEFAULT, plist, H5P_DEFAULT);
  herr_t st   = -1;
  if (set >= 0) st = (d.nx * d.ny * d.nz == 0) ? 0 : H5Dwrite(set, memType, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
  if (set >= 0) H5Dclose(set);
  if (plist != H5P_DEFAULT) H5Pclose(plist);
  H5Sclose(space);
  if (set < 0 || st < 0) throw std::ios_base::failure("Error: cannot write dataset \"" + name + "\"");
}
void Hdf5File::writeMatrix(const std::string& name, const DimensionSizes& dims, const float* data, MatrixDomainType domain)
{
  writeDataset(mFile, name, dims, H5T_IEEE_F32LE, H5T_NATIVE_FLOAT, data, mChunkedOutput, mCompressionLevel);
  writeStringAttribute(name, kMatrixDataTypeName, kDataNames[0]);
  writeStringAttribute(name, kMatrixDomainTypeName, kDomainNames[static_cast<int>(domain)]);
}
void Hdf5File::writeMatrix(const std::string& name, const DimensionSizes& dims, const size_t* data)
{
  writeDataset(mFile, name, dims, H5T_STD_U64LE, H5T_NATIVE_UINT64, data, mChunkedOutput && dims.nElements() > 1,
               mCompressionLevel);
  writeStringAttribute(name, kMatrixDataTypeName, kDataNames[1]);
  writeStringAttribute(name, kMatrixDomainTypeName, kDomainNames[0]);
}
void Hdf5File::remove(const std::string& name)
{
  if (H5Ldelete(mFile, name.c_str(), H5P_DEFAULT) < 0) fail("Error: cannot remove \"" + name + "\" from the file");
}
void Hdf5File::createGroup(const std::string& name)
{
  hid_t g = H5Gcreate2(mFile, name.c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  if (g < 0) fail("Error: cannot create group \"" + name + "\"");
  H5Gclose(g);
}
void Hdf5File::writeCuboid(const std::string& name, const DimensionSizes& d, const float* data)
{
  const int rank = (d.nt > 0) ? 4 : 3;
  hsize_t dims[4] = {d.nt, d.nz, d.ny, d.nx};
  hsize_t* dp = (rank == 4) ? dims : dims + 1;
  if (H5Lexists(mFile, name.c_str(), H5P_DEFAULT) > 0)
  { // an aggregate written by an earlier checkpoint: rewritten in place
    if (getDatasetDimensionSizes(name).nElements() != d.nElements()) fail("Error: cannot rewrite dataset \"" + name + "\"");
    hid_t set = H5Dopen2(mFile, name.c_str(), H5P_DEFAULT);
    const herr_t st = (set < 0) ? -1 : (d.nElements() == 0) ? 0 : H5Dwrite(set, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
    if (set >= 0) H5Dclose(set);
    if (st < 0) fail("Error: cannot rewrite dataset \"" + name + "\"");
    return;
  }
  hid_t space = H5Screate_simple(rank, dp, nullptr);
  hid_t plist = H5P_DEFAULT;
  if (mChunkedOutput && d.nx * d.ny * d.nz > 0)
  { // CuboidOutputStream.cpp:677-690
    constexpr size_t kChunkSize4MB = 1048576;
    hsize_t cdims[4] = {1, d.nz, d.ny, d.nx};
    if (d.nx * d.ny * d.nz > kChunkSize4MB * 8)
    {
      size_t nSlabs = 1;
      while (nSlabs * d.nx * d.ny < kChunkSize4MB) nSlabs++;
      cdims[1] = nSlabs;
    }
    plist = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_chunk(plist, rank, (rank == 4) ? cdims : cdims + 1);
    if (mCompressionLevel > 0) H5Pset_deflate(plist, mCompressionLevel);
  }
  hid_t set = H5Dcreate2(mFile, name.c_str(), H5T_IEEE_F32LE, space, H5P_DEFAULT, plist, H5P_DEFAULT);
  herr_t st = -1;
  if (set >= 0) st = (d.nElements() == 0) ? 0 : H5Dwrite(set, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
  if (set >= 0) H5Dclose(set);
  if (plist != H5P_DEFAULT) H5Pclose(plist);
  H5Sclose(space);
  if (set < 0 || st < 0) fail("Error: cannot write dataset \"" + name + "\"");
  writeStringAttribute(name, kMatrixDataTypeName, kDataNames[0]);
  writeStringAttribute(name, kMatrixDomainTypeName, kDomainNames[0]);
}
void Hdf5File::writeScalarValue(const std::string& name, float value) { writeMatrix(name, DimensionSizes(1, 1, 1), &value, MatrixDomainType::kReal); }
void Hdf5File::writeScalarValue(const std::string& name, size_t value) { writeMatrix(name, DimensionSizes(1, 1, 1), &value); }

void Hdf5File::writeHeader(const std::string& fileType, const std::string& description)
{
  char date[64];
  const time_t now = time(nullptr);
  strftime(date, sizeof(date), "%d/%m/%y, %H:%M:%S", localtime(&now));
  writeStringAttribute("/", "created_by", "kspaceFirstOrder-HIP v0.1 (gfx950)");
  writeStringAttribute("/", "creation_date", date);
  writeStringAttribute("/", "file_description", description);
  writeStringAttribute("/", "file_type", fileType);
  writeStringAttribute("/", "major_version", "1");
  writeStringAttribute("/", "minor_version", "1");
}

Hdf5Input::Hdf5Input(const std::string& fileName)
{
  mFile.open(fileName, true);
  const std::string type = mFile.readFileType();
  if (type != "input") throw std::ios_base::failure("Error: the file \"" + fileName + "\" is not a k-Wave input file (file_type = " + type + ")");
  const std::string major = mFile.readStringAttribute("/", "major_version"), minor = mFile.readStringAttribute("/", "minor_version");
  if (major != "1" || (minor != "0" && minor != "1"))
    throw std::ios_base::failure("Error: unsupported input file version " + major + "." + minor);
}
InputProvider::DataType Hdf5Input::getDatasetType(const std::string& name) const
{
  return mFile.readMatrixDataType(name) == Hdf5File::MatrixDataType::kFloat ? DataType::kFloat : DataType::kLong;
}
