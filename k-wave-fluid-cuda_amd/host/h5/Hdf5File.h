// Hdf5File.h — thin wrapper over the HDF5 C / HL API for the k-Wave file format 1.1.
// Mirror of Hdf5/Hdf5File.{h,cpp} + Hdf5FileHeader.{h,cpp} for what sits on either side of the hot path:
//   * datasets are 3-D, HDF5 dims = (Nz, Ny, Nx); scalars are (1,1,1); complex data is interleaved with a doubled
//     fastest dimension; element types H5T_IEEE_F32LE / H5T_STD_U64LE (Hdf5File.cpp:345-352)
//   * every dataset carries fixed-length string attributes data_type in {"float","long"} and domain_type in
//     {"real","complex"} which readers verify (Hdf5File.cpp:59-68,898-915; RealMatrix.cpp:70-78)
//   * root attributes created_by / creation_date / file_description / file_type / major_version / minor_version
//     (Hdf5FileHeader.cpp:62-87)
// Optional component: compiled only into libkwave_host_h5.so (needs libhdf5 + libhdf5_hl).
#ifndef KW_HOST_HDF5_FILE_H
#define KW_HOST_HDF5_FILE_H
#include <hdf5.h>
#include <hdf5_hl.h>

#include <string>
#include <vector>

#include "../DimensionSizes.h"
#include "../InputProvider.h"

class Hdf5File
{
 public:
  enum class MatrixDataType { kFloat = 0, kLong = 1 };
  enum class MatrixDomainType { kReal = 0, kComplex = 1 };
  static const std::string kMatrixDomainTypeName, kMatrixDataTypeName;

  Hdf5File() = default;
  ~Hdf5File();
  void create(const std::string& fileName);                       // Hdf5File.cpp:97-118 (truncate)
  void open(const std::string& fileName, bool readOnly = true);   // :126-146
  bool isOpen() const { return mFile >= 0; }
  hid_t handle() const { return mFile; }
  void close();
  bool datasetExists(const std::string& name) const;
  DimensionSizes getDatasetDimensionSizes(const std::string& name) const; // returned as (x,y,z) = HDF5 dims reversed
  size_t getDatasetSize(const std::string& name) const;
  MatrixDataType   readMatrixDataType(const std::string& name) const;
  MatrixDomainType readMatrixDomainType(const std::string& name) const;
  void readCompleteDataset(const std::string& name, size_t nElements, float* data) const;  // :791-803
  /// planes [z0, z0 + nPlanes) of a 3-D float dataset (readHyperSlab, :817-870) — what one rank of a slab-decomposed
  /// run needs of a grid-sized input array
  void readPlanes(const std::string& name, size_t z0, size_t nPlanes, float* data) const;
  void readCompleteDataset(const std::string& name, size_t nElements, size_t* data) const; // :805-815
  /// write a whole 3-D dataset (x,y,z sizes) + its data_type / domain_type attributes; a dataset of that name and extent
  /// that exists already is rewritten in place
  void writeMatrix(const std::string& name, const DimensionSizes& dims, const float* data, MatrixDomainType domain);
  void writeMatrix(const std::string& name, const DimensionSizes& dims, const size_t* data);
  /// group + dataset of one sampled cuboid (CuboidOutputStream.cpp:656-722): dims = (nx, ny, nz[, nt]); 4-D when nt > 0;
  /// chunk = one time step of the cuboid, cut into z-slabs of ~4 MB above 32 MB
  void createGroup(const std::string& name);
  void remove(const std::string& name); ///< unlink a dataset (its space is not reclaimed until the file is repacked)
  void writeCuboid(const std::string& name, const DimensionSizes& dims, const float* data);
  void writeScalarValue(const std::string& name, float value);
  void writeScalarValue(const std::string& name, size_t value);
  void writeStringAttribute(const std::string& dataset, const std::string& attr, const std::string& value); // "/" = root
  void writeLongLongAttribute(const std::string& dataset, const std::string& attr, long long value);
  void writeFloatAttribute(const std::string& dataset, const std::string& attr, float value);
  double readNumericAttribute(const std::string& dataset, const std::string& attr) const; // integer or float attribute
  std::string readStringAttribute(const std::string& dataset, const std::string& attr) const;
  /// root header attributes (Hdf5FileHeader.cpp:126-149 write, :155-200 read/check)
  void writeHeader(const std::string& fileType, const std::string& description);
  std::string readFileType() const { return readStringAttribute("/", "file_type"); }
  /// datasets written from now on are chunked like the reference's output (RealMatrix.cpp:88-121) and deflated at
  /// `compressionLevel` (0-9; the reference's -c, default 0)
  void setOutputLayout(bool chunked, unsigned compressionLevel) { mChunkedOutput = chunked; mCompressionLevel = compressionLevel; }

 private:
  hid_t mFile = -1;
  std::string mName;
  bool     mChunkedOutput    = false;
  unsigned mCompressionLevel = 0;
};

/// InputProvider backed by an HDF5 input file (what Parameters::readScalarsFromInputFile and the matrices read)
class Hdf5Input : public InputProvider
{
 public:
  explicit Hdf5Input(const std::string& fileName);
  bool datasetExists(const std::string& name) const override { return mFile.datasetExists(name); }
  DimensionSizes getDatasetDimensionSizes(const std::string& name) const override { return mFile.getDatasetDimensionSizes(name); }
  DataType getDatasetType(const std::string& name) const override;
  void readFloat(const std::string& name, float* dst, size_t n) const override { mFile.readCompleteDataset(name, n, dst); }
  void readIndex(const std::string& name, size_t* dst, size_t n) const override { mFile.readCompleteDataset(name, n, dst); }

 private:
  Hdf5File mFile;
};
#endif
