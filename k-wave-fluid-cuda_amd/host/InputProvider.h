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

/// 2-D simulations (Nz == 1; the reference's SD::k2D instantiations): a 2-D input file has no z datasets
/// (MatrixContainer.cpp:102-200 creates them only for 3-D).  The solver here runs the 3-D code with one plane — the
/// z-gradient operators are zero, so u_z and rho_z stay zero, the 3-D FFT of one plane is the 2-D FFT, and
/// fftDivider = 1/(Nx*Ny) — and this adapter supplies the z datasets that make that exact: ddz_k_shift_* = 0,
/// pml_z = pml_z_sgz = 1, z_shift_neg_r = 1, rho0_sgz = rho0_sgx.  What is genuinely 2-D in the arithmetic (the initial
/// density split by 2, sources not written to rho_z) is handled where the reference handles it (kw_add_*_source).
class Input2DAdapter : public InputProvider
{
 public:
  explicit Input2DAdapter(const InputProvider& in) : mIn(in) {}
  bool datasetExists(const std::string& name) const override { return mIn.datasetExists(name) || isSynth(name); }
  DimensionSizes getDatasetDimensionSizes(const std::string& name) const override
  {
    if (mIn.datasetExists(name)) return mIn.getDatasetDimensionSizes(name);
    if (name == "ddz_k_shift_pos" || name == "ddz_k_shift_neg" || name == "z_shift_neg_r") return DimensionSizes(2, 1, 1);
    if (name == "pml_z" || name == "pml_z_sgz") return DimensionSizes(1, 1, 1);
    if (name == "rho0_sgz") return mIn.getDatasetDimensionSizes("rho0_sgx");
    throw std::runtime_error("Dataset " + name + " not found in the input");
  }
  DataType getDatasetType(const std::string& name) const override
  {
    return mIn.datasetExists(name) ? mIn.getDatasetType(name) : DataType::kFloat;
  }
  void readFloat(const std::string& name, float* dst, size_t n) const override
  {
    if (mIn.datasetExists(name)) { mIn.readFloat(name, dst, n); return; }
    if (name == "rho0_sgz") { mIn.readFloat("rho0_sgx", dst, n); return; }
    if (!isSynth(name) || n != getDatasetDimensionSizes(name).nElements())
      throw std::runtime_error("Dataset " + name + " not found in the input");
    if (name == "pml_z" || name == "pml_z_sgz") dst[0] = 1.0f;
    else if (name == "z_shift_neg_r") { dst[0] = 1.0f; dst[1] = 0.0f; }
    else { dst[0] = 0.0f; dst[1] = 0.0f; }
  }
  void readIndex(const std::string& name, size_t* dst, size_t n) const override { mIn.readIndex(name, dst, n); }

 private:
  static bool isSynth(const std::string& n)
  {
    return n == "ddz_k_shift_pos" || n == "ddz_k_shift_neg" || n == "z_shift_neg_r" || n == "pml_z" ||
           n == "pml_z_sgz" || n == "rho0_sgz";
  }
  const InputProvider& mIn;
};
#endif
