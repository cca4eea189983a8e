// MatrixContainer.cpp — see MatrixContainer.h.  Which matrices exist when follows Containers/MatrixContainer.cpp:73-411 in
// *content*; here it is a schema table (kSchema below) walked by init().
#include "MatrixContainer.h"

#include "MatrixNames.h"
#include "Parameters.h"

// ---- which matrices exist for a given problem: one table -------------------------------------------------------------
// The reference decides this in 340 lines of mContainer[...].set(...) statements under nested conditions
// (Containers/MatrixContainer.cpp:73-411); the content is the same here — enum, element type, extent, "read from the input
// file?", "part of a checkpoint?", dataset name — but held as rows of a schema, each with the predicate that admits it.
namespace
{
using MT = MatrixRecord::MatrixType;
using MI = MatrixContainer::MatrixIdx;
using P  = Parameters;

// extent of a matrix in terms of the problem
enum class Extent
{
  kGrid,        // Nx x Ny x Nz (the local slab in a slab run)
  kSpectrum,    // Nx/2+1 x Ny x Nz
  kAlongX, kAlongY, kAlongZ,          // 1-D over an axis of the local grid
  kSpectrumX, kSpectrumY,             // 1-D operator over an axis of the reduced grid
  kGlobalZ,                           // 1-D over the global z axis (z operators are not cut by a slab decomposition)
  kHalfX, kHalfY, kHalfGlobalZ,       // N/2+1 shift vectors of the non-staggered velocity
  kByRule                             // computed by the row's own rule
};
DimensionSizes extentOf(Extent e, const P& p)
{
  const DimensionSizes g = p.getFullDimensionSizes(), r = p.getReducedDimensionSizes();
  const size_t nzGlobal = p.getGlobalDimensionSizes().nz;
  switch (e)
  {
    case Extent::kGrid: return g;
    case Extent::kSpectrum: return r;
    case Extent::kAlongX: return DimensionSizes(g.nx, 1, 1);
    case Extent::kAlongY: return DimensionSizes(1, g.ny, 1);
    case Extent::kAlongZ: return DimensionSizes(1, 1, g.nz);
    case Extent::kSpectrumX: return DimensionSizes(r.nx, 1, 1);
    case Extent::kSpectrumY: return DimensionSizes(1, r.ny, 1);
    case Extent::kGlobalZ: return DimensionSizes(1, 1, nzGlobal);
    case Extent::kHalfX: return DimensionSizes(g.nx / 2 + 1, 1, 1);
    case Extent::kHalfY: return DimensionSizes(1, g.ny / 2 + 1, 1);
    case Extent::kHalfGlobalZ: return DimensionSizes(1, 1, nzGlobal / 2 + 1);
    default: return DimensionSizes();
  }
}

// predicates over the parameters (what the medium, the sources and the requested outputs need)
bool always(const P&) { return true; }
bool c0Array(const P& p) { return !p.getC0ScalarFlag(); }
bool rho0Array(const P& p) { return !p.getRho0ScalarFlag(); }
bool nonUniform(const P& p) { return p.getNonUniformGridFlag() != 0; }
bool nonUniformScalarRho0(const P& p) { return nonUniform(p) && p.getRho0ScalarFlag(); }
bool bOnAArray(const P& p) { return p.getNonLinearFlag() && !p.getBOnAScalarFlag(); }
bool absorbing(const P& p) { return p.getAbsorbingFlag() != 0; }
bool absorbingArrays(const P& p) { return absorbing(p) && !(p.getC0ScalarFlag() && p.getAlphaCoeffScalarFlag()); }
bool alphaCoeffArray(const P& p) { return absorbing(p) && !p.getAlphaCoeffScalarFlag(); }
bool noAlphaCoeffArray(const P& p) { return !alphaCoeffArray(p); }
bool indexMask(const P& p) { return p.getSensorMaskType() == P::SensorMaskType::kIndex && p.getSensorMaskIndexSize() > 0; }
bool cornersMask(const P& p) { return p.getSensorMaskType() == P::SensorMaskType::kCorners; }
bool p0Source(const P& p) { return p.getInitialPressureSourceFlag() == 1; }
bool transducer(const P& p) { return p.getTransducerSourceFlag() != 0; }
bool uxSource(const P& p) { return p.getVelocityXSourceFlag() != 0; }
bool uySource(const P& p) { return p.getVelocityYSourceFlag() != 0; }
bool uzSource(const P& p) { return p.getVelocityZSourceFlag() != 0; }
bool anyVelocityIndex(const P& p) { return transducer(p) || uxSource(p) || uySource(p) || uzSource(p); }
bool pSource(const P& p) { return p.getPressureSourceFlag() != 0; }
bool kSpaceCorrectedSource(const P& p)
{
  return ((p.getVelocitySourceMode() == P::SourceMode::kAdditive) || (p.getPressureSourceMode() == P::SourceMode::kAdditive)) &&
         (pSource(p) || uxSource(p) || uySource(p) || uzSource(p));
}
bool shifted(const P& p) { return p.needsShiftedVelocity(); }
bool shifted3D(const P& p) { return shifted(p) && p.isSimulation3D(); }

// extents that follow a rule of their own
DimensionSizes seriesOf(size_t many, size_t indexSize, size_t steps) { return (many == 0) ? DimensionSizes(1, 1, steps) : DimensionSizes(1, indexSize, steps); }
DimensionSizes dimsIndexMask(const P& p) { return DimensionSizes(p.getSensorMaskIndexSize(), 1, 1); }
DimensionSizes dimsCornersMask(const P& p) { return DimensionSizes(6, p.getSensorMaskCornersSize(), 1); }
DimensionSizes dimsVelocityIndex(const P& p) { return DimensionSizes(1, 1, p.getVelocitySourceIndexSize()); }
DimensionSizes dimsTransducerInput(const P& p) { return DimensionSizes(1, 1, p.getTransducerSourceInputSize()); }
DimensionSizes dimsPressureIndex(const P& p) { return DimensionSizes(1, 1, p.getPressureSourceIndexSize()); }
DimensionSizes dimsPressureInput(const P& p) { return seriesOf(p.getPressureSourceMany(), p.getPressureSourceIndexSize(), p.getPressureSourceFlag()); }
DimensionSizes dimsUxInput(const P& p) { return seriesOf(p.getVelocitySourceMany(), p.getVelocitySourceIndexSize(), p.getVelocityXSourceFlag()); }
DimensionSizes dimsUyInput(const P& p) { return seriesOf(p.getVelocitySourceMany(), p.getVelocitySourceIndexSize(), p.getVelocityYSourceFlag()); }
DimensionSizes dimsUzInput(const P& p) { return seriesOf(p.getVelocitySourceMany(), p.getVelocitySourceIndexSize(), p.getVelocityZSourceFlag()); }
DimensionSizes dimsShiftTemp(const P& p)
{ // the 1-D transform workspace of the non-staggered velocity: the largest of the three half-spectra (MatrixContainer.cpp:336-352)
  const DimensionSizes g = p.getFullDimensionSizes();
  const size_t hx = g.nx / 2 + 1, hy = g.ny / 2 + 1, hz = g.nz / 2 + 1;
  const size_t cutX = hx * g.ny * g.nz, cutY = g.nx * hy * g.nz, cutZ = g.nx * g.ny * hz;
  DimensionSizes d = g;
  if (cutX >= cutY && cutX >= cutZ) d.nx = hx;
  else if (cutY >= cutX && cutY >= cutZ) d.ny = hy;
  else d.nz = hz;
  return d;
}

enum : unsigned { kFromFile = 1u, kInCheckpoint = 2u };
struct Row
{
  MI          idx;
  MT          type;
  Extent      extent;
  DimensionSizes (*rule)(const P&); // Extent::kByRule
  unsigned    flags;
  std::string name;                 // dataset name in the input / checkpoint file, or a label for matrices that are in neither
  bool      (*admit)(const P&);
};
#define GRID(idx, flags, name, when) { MI::idx, MT::kReal, Extent::kGrid, nullptr, flags, name, when }

const Row kSchema[] = {
  // medium and k-space operators
  { MI::kKappa, MT::kReal, Extent::kSpectrum, nullptr, 0, "kappa_r", always },
  GRID(kC2, kFromFile, kC0Name, c0Array), // c0 is read into the c^2 matrix and squared in pre-processing
  GRID(kRho0, kFromFile, kRho0Name, rho0Array),
  // read as rho0_sg*, turned into dt / rho0_sg* in pre-processing (KSpaceFirstOrderSolver.cpp:825-830)
  GRID(kDtRho0Sgx, kFromFile, kRho0SgxName, rho0Array),
  GRID(kDtRho0Sgy, kFromFile, kRho0SgyName, rho0Array),
  GRID(kDtRho0Sgz, kFromFile, kRho0SgzName, rho0Array),
  // homogeneous density on a non-uniform grid: dt / rho0_sg * d?ud?n_sg? as per-voxel arrays built in pre-processing, so
  // that the velocity kernels need no third variant (the reference has one: SolverCudaKernels.cu:372-410)
  GRID(kDtRho0Sgx, 0, "dt_rho0_sgx_nonuniform", nonUniformScalarRho0),
  GRID(kDtRho0Sgy, 0, "dt_rho0_sgy_nonuniform", nonUniformScalarRho0),
  GRID(kDtRho0Sgz, 0, "dt_rho0_sgz_nonuniform", nonUniformScalarRho0),
  GRID(kBOnA, kFromFile, kBonAName, bOnAArray),
  GRID(kAbsorbTau, 0, "absorb_tau", absorbingArrays),
  GRID(kAbsorbEta, 0, "absorb_eta", absorbingArrays),
  { MI::kAbsorbNabla1, MT::kReal, Extent::kSpectrum, nullptr, 0, "absorb_nabla1_r", absorbing },
  { MI::kAbsorbNabla2, MT::kReal, Extent::kSpectrum, nullptr, 0, "absorb_nabla2_r", absorbing },
  { MI::kSourceKappa, MT::kReal, Extent::kSpectrum, nullptr, 0, "source_kappa_r", kSpaceCorrectedSource },
  { MI::kDdxKShiftPosR, MT::kComplex, Extent::kSpectrumX, nullptr, kFromFile, kDdxKShiftPosRName, always },
  { MI::kDdyKShiftPos, MT::kComplex, Extent::kSpectrumY, nullptr, kFromFile, kDdyKShiftPosName, always },
  { MI::kDdzKShiftPos, MT::kComplex, Extent::kGlobalZ, nullptr, kFromFile, kDdzKShiftPosName, always },
  { MI::kDdxKShiftNegR, MT::kComplex, Extent::kSpectrumX, nullptr, kFromFile, kDdxKShiftNegRName, always },
  { MI::kDdyKShiftNeg, MT::kComplex, Extent::kSpectrumY, nullptr, kFromFile, kDdyKShiftNegName, always },
  { MI::kDdzKShiftNeg, MT::kComplex, Extent::kGlobalZ, nullptr, kFromFile, kDdzKShiftNegName, always },
  { MI::kPmlXSgx, MT::kReal, Extent::kAlongX, nullptr, kFromFile, kPmlXSgxName, always },
  { MI::kPmlYSgy, MT::kReal, Extent::kAlongY, nullptr, kFromFile, kPmlYSgyName, always },
  { MI::kPmlZSgz, MT::kReal, Extent::kAlongZ, nullptr, kFromFile, kPmlZSgzName, always },
  { MI::kPmlX, MT::kReal, Extent::kAlongX, nullptr, kFromFile, kPmlXName, always },
  { MI::kPmlY, MT::kReal, Extent::kAlongY, nullptr, kFromFile, kPmlYName, always },
  { MI::kPmlZ, MT::kReal, Extent::kAlongZ, nullptr, kFromFile, kPmlZName, always },
  // non-uniform grid scalings (MatrixContainer.cpp:301-329)
  { MI::kDxudxn, MT::kReal, Extent::kAlongX, nullptr, kFromFile, kDxudxnName, nonUniform },
  { MI::kDyudyn, MT::kReal, Extent::kAlongY, nullptr, kFromFile, kDyudynName, nonUniform },
  { MI::kDzudzn, MT::kReal, Extent::kAlongZ, nullptr, kFromFile, kDzudznName, nonUniform },
  { MI::kDxudxnSgx, MT::kReal, Extent::kAlongX, nullptr, kFromFile, kDxudxnSgxName, nonUniform },
  { MI::kDyudynSgy, MT::kReal, Extent::kAlongY, nullptr, kFromFile, kDyudynSgyName, nonUniform },
  { MI::kDzudznSgz, MT::kReal, Extent::kAlongZ, nullptr, kFromFile, kDzudznSgzName, nonUniform },
  // state: what a checkpoint keeps (MatrixContainer.cpp:504-537)
  GRID(kP, kInCheckpoint, kPName, always),
  GRID(kRhoX, kInCheckpoint, kRhoXName, always),
  GRID(kRhoY, kInCheckpoint, kRhoYName, always),
  GRID(kRhoZ, kInCheckpoint, kRhoZName, always),
  GRID(kUxSgx, kInCheckpoint, kUxSgxName, always),
  GRID(kUySgy, kInCheckpoint, kUySgyName, always),
  GRID(kUzSgz, kInCheckpoint, kUzSgzName, always),
  GRID(kDuxdx, 0, "duxdx", always),
  GRID(kDuydy, 0, "duydy", always),
  GRID(kDuzdz, 0, "duzdz", always),
  // sensor masks and sources (MatrixContainer.cpp:203-300)
  { MI::kSensorMaskIndex, MT::kIndex, Extent::kByRule, dimsIndexMask, kFromFile, kSensorMaskIndexName, indexMask },
  { MI::kSensorMaskCorners, MT::kIndex, Extent::kByRule, dimsCornersMask, kFromFile, kSensorMaskCornersName, cornersMask },
  GRID(kInitialPressureSourceInput, kFromFile, kInitialPressureSourceInputName, p0Source),
  { MI::kVelocitySourceIndex, MT::kIndex, Extent::kByRule, dimsVelocityIndex, kFromFile, kVelocitySourceIndexName, anyVelocityIndex },
  { MI::kDelayMask, MT::kIndex, Extent::kByRule, dimsVelocityIndex, kFromFile, kDelayMaskName, transducer },
  { MI::kTransducerSourceInput, MT::kReal, Extent::kByRule, dimsTransducerInput, kFromFile, kTransducerSourceInputName, transducer },
  { MI::kPressureSourceInput, MT::kReal, Extent::kByRule, dimsPressureInput, kFromFile, kPressureSourceInputName, pSource },
  { MI::kPressureSourceIndex, MT::kIndex, Extent::kByRule, dimsPressureIndex, kFromFile, kPressureSourceIndexName, pSource },
  { MI::kVelocityXSourceInput, MT::kReal, Extent::kByRule, dimsUxInput, kFromFile, kVelocityXSourceInputName, uxSource },
  { MI::kVelocityYSourceInput, MT::kReal, Extent::kByRule, dimsUyInput, kFromFile, kVelocityYSourceInputName, uySource },
  { MI::kVelocityZSourceInput, MT::kReal, Extent::kByRule, dimsUzInput, kFromFile, kVelocityZSourceInputName, uzSource },
  // non-staggered velocity (MatrixContainer.cpp:330-385); z lines and z_shift_neg_r keep their global length on a slab
  { MI::kTempHipFftShift, MT::kFft, Extent::kByRule, dimsShiftTemp, 0, "hipfft_shift_temp", shifted },
  GRID(kUxShifted, 0, "ux_shifted", shifted),
  GRID(kUyShifted, 0, "uy_shifted", shifted),
  GRID(kUzShifted, 0, "uz_shifted", shifted3D),
  { MI::kXShiftNegR, MT::kComplex, Extent::kHalfX, nullptr, kFromFile, kXShiftNegRName, shifted },
  { MI::kYShiftNegR, MT::kComplex, Extent::kHalfY, nullptr, kFromFile, kYShiftNegRName, shifted },
  { MI::kZShiftNegR, MT::kComplex, Extent::kHalfGlobalZ, nullptr, kFromFile, kZShiftNegRName, shifted3D },
  // temporaries (MatrixContainer.cpp:387-410): alpha_coeff is read *into* Temp1 and consumed by the generators
  GRID(kTemp1RealND, kFromFile, kAlphaCoeffName, alphaCoeffArray),
  GRID(kTemp1RealND, 0, "Temp_1_RS3D", noAlphaCoeffArray),
  GRID(kTemp2RealND, 0, "Temp_2_RS3D", always),
  GRID(kTemp3RealND, 0, "Temp_3_RS3D", always),
  { MI::kTempHipFftX, MT::kFft, Extent::kSpectrum, nullptr, 0, "hipfft_X_temp", always },
  { MI::kTempHipFftY, MT::kFft, Extent::kSpectrum, nullptr, 0, "hipfft_Y_temp", always },
  { MI::kTempHipFftZ, MT::kFft, Extent::kSpectrum, nullptr, 0, "hipfft_Z_temp", always },
};
#undef GRID
} // namespace

void MatrixContainer::init()
{
  const Parameters& params = Parameters::getInstance();
  for (const Row& row : kSchema)
  {
    if (!row.admit(params)) continue;
    const DimensionSizes dims = (row.extent == Extent::kByRule) ? row.rule(params) : extentOf(row.extent, params);
    mContainer[row.idx].set(row.type, dims, (row.flags & kFromFile) != 0, (row.flags & kInCheckpoint) != 0, row.name);
  }
}

void MatrixContainer::createMatrices()
{
  using MT = MatrixRecord::MatrixType;
  for (auto& it : mContainer)
  {
    if (it.second.matrixPtr != nullptr) throw std::invalid_argument("Error: Matrix [" + it.second.matrixName + "] was reallocated");
    switch (it.second.matrixType)
    {
      case MT::kReal: it.second.matrixPtr = new RealMatrix(it.second.dimensionSizes); break;
      case MT::kComplex: it.second.matrixPtr = new ComplexMatrix(it.second.dimensionSizes); break;
      case MT::kIndex: it.second.matrixPtr = new IndexMatrix(it.second.dimensionSizes); break;
      case MT::kFft: it.second.matrixPtr = new HipFftComplexMatrix(it.second.dimensionSizes); break;
    }
  }
}

void MatrixContainer::freeMatrices()
{
  for (auto& it : mContainer)
  {
    delete it.second.matrixPtr;
    it.second.matrixPtr = nullptr;
  }
  mContainer.clear();
}

void MatrixContainer::loadDataFromInputFile(const InputProvider& in)
{
  for (auto& it : mContainer)
    if (it.second.loadData) it.second.matrixPtr->readData(in, it.second.matrixName);
}

void MatrixContainer::copyMatricesToDevice()
{
  for (auto& it : mContainer) it.second.matrixPtr->copyToDevice();
}
void MatrixContainer::copyMatricesFromDevice()
{
  for (auto& it : mContainer) it.second.matrixPtr->copyFromDevice();
}
