// InputProvider.h — where the solver's input datasets come from.
// The reference reads them from an HDF5 file (Hdf5/Hdf5File.cpp:767-815); here the same named datasets can come
// from memory (MemoryInput: used by bench.py / tests, and on boxes without HDF5) or from a file (Hdf5Input,
// host/Hdf5Input.cpp, optional component).  Dataset names and shapes are the file-format-1.1 contract.
#ifndef KW_HOST_INPUT_PROVIDER_H
#define KW_HOST_INPUT_PROVIDER_H
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>

#include "DimensionSizes.h"

class InputProvider
{
 public:
  enum class DataType { kFloat, kLong };
  virtual ~InputProvider() = default;
  virtual bool           datasetExists(const std::string& name) const = 0;
  virtual DimensionSizes getDatasetDimensionSizes(const std::string& name) const = 0;
  virtual DataType       getDatasetType(const std::string& name) const = 0;
  /// read n elements; throws std::ios_base::failure-like runtime_error on size/type mismatch (RealMatrix.cpp:70-78)
  virtual void readFloat(const std::string& name, float* dst, size_t n) const = 0;
  virtual void readIndex(const std::string& name, size_t* dst, size_t n) const = 0;

  size_t getDatasetSize(const std::string& name) const { return getDatasetDimensionSizes(name).nElements(); }
  void   readScalarValue(const std::string& name, float& v) const { readFloat(name, &v, 1); }
  void   readScalarValue(const std::string& name, size_t& v) const { readIndex(name, &v, 1); }
};

class MemoryInput : public InputProvider
{
 public:
  struct Entry { const void* data; DataType type; DimensionSizes dims; };
  void add(const std::string& name, const void* data, DataType type, const DimensionSizes& dims)
  {
    mEntries[name] = Entry{data, type, dims};
  }
  bool datasetExists(const std::string& name) const override { return mEntries.count(name) != 0; }
  DimensionSizes getDatasetDimensionSizes(const std::string& name) const override { return get(name).dims; }
  DataType getDatasetType(const std::string& name) const override { return get(name).type; }
  void readFloat(const std::string& name, float* dst, size_t n) const override
  {
    const Entry& e = get(name);
    if (e.type != DataType::kFloat) throw std::runtime_error("Dataset " + name + " has wrong data type (expected float)");
    if (e.dims.nElements() != n) throw std::runtime_error("Dataset " + name + " has wrong dimension sizes");
    std::memcpy(dst, e.data, n * sizeof(float));
  }
  void readIndex(const std::string& name, size_t* dst, size_t n) const override
  {
    const Entry& e = get(name);
    if (e.type != DataType::kLong) throw std::runtime_error("Dataset " + name + " has wrong data type (expected long)");
    if (e.dims.nElements() != n) throw std::runtime_error("Dataset " + name + " has wrong dimension sizes");
    std::memcpy(dst, e.data, n * sizeof(uint64_t));
  }

 private:
  const Entry& get(const std::string& name) const
  {
    auto it = mEntries.find(name);
    if (it == mEntries.end()) throw std::runtime_error("Dataset " + name + " not found in the input");
    return it->second;
  }
  std::map<std::string, Entry> mEntries;
};
#endif
