// MatrixContainer.cpp — see MatrixContainer.h.  The set of matrices follows Containers/MatrixContainer.cpp:73-411
// line by line in *content* (which arrays exist when), restated for the 3-D uniform-grid scope of this build.
#include "MatrixContainer.h"

#include "MatrixNames.h"
#include "Parameters.h"

void MatrixContainer::init()
{
  using MT = MatrixRecord::MatrixType;
  using MI = MatrixContainer::MatrixIdx;
  const Parameters& params = Parameters::getInstance();
  const DimensionSizes fullDims = params.getFullDimensionSizes(), reducedDims = params.getReducedDimensionSizes();
  const size_t nzGlobal = params.getGlobalDimensionSizes().nz; // 1-D z operators stay global on a slab
  constexpr bool kLoad = true, kNoLoad = false, kCheckpoint = true, kNoCheckpoint = false;

  mContainer[MI::kKappa].set(MT::kReal, reducedDims, kNoLoad, kNoCheckpoint, "kappa_r");
  if (!params.getC0ScalarFlag()) mContainer[MI::kC2].set(MT::kReal, fullDims, kLoad, kNoCheckpoint, kC0Name);
  mContainer[MI::kP].set(MT::kReal, fullDims, kNoLoad, kCheckpoint, kPName);
  mContainer[MI::kRhoX].set(MT::kReal, fullDims, kNoLoad, kCheckpoint, kRhoXName);
  mContainer[MI::kRhoY].set(MT::kReal, fullDims, kNoLoad, kCheckpoint, kRhoYName);
  mContainer[MI::kRhoZ].set(MT::kReal, fullDims, kNoLoad, kCheckpoint, kRhoZName);
  mContainer[MI::kUxSgx].set(MT::kReal, fullDims, kNoLoad, kCheckpoint, kUxSgxName);
  mContainer[MI::kUySgy].set(MT::kReal, fullDims, kNoLoad, kCheckpoint, kUySgyName);
  mContainer[MI::kUzSgz].set(MT::kReal, fullDims, kNoLoad, kCheckpoint, kUzSgzName);
  mContainer[MI::kDuxdx].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "duxdx");
  mContainer[MI::kDuydy].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "duydy");
  mContainer[MI::kDuzdz].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "duzdz");
  if (params.getNonUniformGridFlag() != 0)
  { // MatrixContainer.cpp:301-329
    mContainer[MI::kDxudxn].set(MT::kReal, DimensionSizes(fullDims.nx, 1, 1), kLoad, kNoCheckpoint, kDxudxnName);
    mContainer[MI::kDyudyn].set(MT::kReal, DimensionSizes(1, fullDims.ny, 1), kLoad, kNoCheckpoint, kDyudynName);
    mContainer[MI::kDzudzn].set(MT::kReal, DimensionSizes(1, 1, fullDims.nz), kLoad, kNoCheckpoint, kDzudznName);
    mContainer[MI::kDxudxnSgx].set(MT::kReal, DimensionSizes(fullDims.nx, 1, 1), kLoad, kNoCheckpoint, kDxudxnSgxName);
    mContainer[MI::kDyudynSgy].set(MT::kReal, DimensionSizes(1, fullDims.ny, 1), kLoad, kNoCheckpoint, kDyudynSgyName);
    mContainer[MI::kDzudznSgz].set(MT::kReal, DimensionSizes(1, 1, fullDims.nz), kLoad, kNoCheckpoint, kDzudznSgzName);
    if (params.getRho0ScalarFlag())
    { // homogeneous density on a non-uniform grid: dt/rho0_sg * d?ud?n_sg? as per-voxel arrays, built in pre-processing,
      // so that the velocity kernels need no third variant (the reference has one: SolverCudaKernels.cu:372-410)
      mContainer[MI::kDtRho0Sgx].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "dt_rho0_sgx_nonuniform");
      mContainer[MI::kDtRho0Sgy].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "dt_rho0_sgy_nonuniform");
      mContainer[MI::kDtRho0Sgz].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "dt_rho0_sgz_nonuniform");
    }
  }
  if (!params.getRho0ScalarFlag())
  {
    mContainer[MI::kRho0].set(MT::kReal, fullDims, kLoad, kNoCheckpoint, kRho0Name);
    // loaded as rho0_sg*, turned into dt/rho0_sg* by preProcessing (KSpaceFirstOrderSolver.cpp:825-830)
    mContainer[MI::kDtRho0Sgx].set(MT::kReal, fullDims, kLoad, kNoCheckpoint, kRho0SgxName);
    mContainer[MI::kDtRho0Sgy].set(MT::kReal, fullDims, kLoad, kNoCheckpoint, kRho0SgyName);
    mContainer[MI::kDtRho0Sgz].set(MT::kReal, fullDims, kLoad, kNoCheckpoint, kRho0SgzName);
  }
  mContainer[MI::kDdxKShiftPosR].set(MT::kComplex, DimensionSizes(reducedDims.nx, 1, 1), kLoad, kNoCheckpoint, kDdxKShiftPosRName);
  mContainer[MI::kDdyKShiftPos].set(MT::kComplex, DimensionSizes(1, reducedDims.ny, 1), kLoad, kNoCheckpoint, kDdyKShiftPosName);
  mContainer[MI::kDdzKShiftPos].set(MT::kComplex, DimensionSizes(1, 1, nzGlobal), kLoad, kNoCheckpoint, kDdzKShiftPosName);
  mContainer[MI::kDdxKShiftNegR].set(MT::kComplex, DimensionSizes(reducedDims.nx, 1, 1), kLoad, kNoCheckpoint, kDdxKShiftNegRName);
  mContainer[MI::kDdyKShiftNeg].set(MT::kComplex, DimensionSizes(1, reducedDims.ny, 1), kLoad, kNoCheckpoint, kDdyKShiftNegName);
  mContainer[MI::kDdzKShiftNeg].set(MT::kComplex, DimensionSizes(1, 1, nzGlobal), kLoad, kNoCheckpoint, kDdzKShiftNegName);
  mContainer[MI::kPmlXSgx].set(MT::kReal, DimensionSizes(fullDims.nx, 1, 1), kLoad, kNoCheckpoint, kPmlXSgxName);
  mContainer[MI::kPmlYSgy].set(MT::kReal, DimensionSizes(1, fullDims.ny, 1), kLoad, kNoCheckpoint, kPmlYSgyName);
  mContainer[MI::kPmlZSgz].set(MT::kReal, DimensionSizes(1, 1, fullDims.nz), kLoad, kNoCheckpoint, kPmlZSgzName);
  mContainer[MI::kPmlX].set(MT::kReal, DimensionSizes(fullDims.nx, 1, 1), kLoad, kNoCheckpoint, kPmlXName);
  mContainer[MI::kPmlY].set(MT::kReal, DimensionSizes(1, fullDims.ny, 1), kLoad, kNoCheckpoint, kPmlYName);
  mContainer[MI::kPmlZ].set(MT::kReal, DimensionSizes(1, 1, fullDims.nz), kLoad, kNoCheckpoint, kPmlZName);
  if (params.getNonLinearFlag() && !params.getBOnAScalarFlag())
    mContainer[MI::kBOnA].set(MT::kReal, fullDims, kLoad, kNoCheckpoint, kBonAName);
  if (params.getAbsorbingFlag() != 0)
  {
    if (!((params.getC0ScalarFlag()) && (params.getAlphaCoeffScalarFlag())))
    {
      mContainer[MI::kAbsorbTau].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "absorb_tau");
      mContainer[MI::kAbsorbEta].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "absorb_eta");
    }
    mContainer[MI::kAbsorbNabla1].set(MT::kReal, reducedDims, kNoLoad, kNoCheckpoint, "absorb_nabla1_r");
    mContainer[MI::kAbsorbNabla2].set(MT::kReal, reducedDims, kNoLoad, kNoCheckpoint, "absorb_nabla2_r");
  }
  if (params.getSensorMaskType() == Parameters::SensorMaskType::kIndex && params.getSensorMaskIndexSize() > 0)
    mContainer[MI::kSensorMaskIndex].set(MT::kIndex, DimensionSizes(params.getSensorMaskIndexSize(), 1, 1), kLoad, kNoCheckpoint, kSensorMaskIndexName);
  if (params.getSensorMaskType() == Parameters::SensorMaskType::kCorners)
    mContainer[MI::kSensorMaskCorners].set(MT::kIndex, DimensionSizes(6, params.getSensorMaskCornersSize(), 1), kLoad, kNoCheckpoint, kSensorMaskCornersName);

  // ---- sources (MatrixContainer.cpp:203-300) ----
  if (params.getInitialPressureSourceFlag() == 1)
    mContainer[MI::kInitialPressureSourceInput].set(MT::kReal, fullDims, kLoad, kNoCheckpoint, kInitialPressureSourceInputName);
  if ((params.getTransducerSourceFlag() != 0) || (params.getVelocityXSourceFlag() != 0) ||
      (params.getVelocityYSourceFlag() != 0) || (params.getVelocityZSourceFlag() != 0))
    mContainer[MI::kVelocitySourceIndex].set(MT::kIndex, DimensionSizes(1, 1, params.getVelocitySourceIndexSize()), kLoad, kNoCheckpoint, kVelocitySourceIndexName);
  if (params.getTransducerSourceFlag() != 0)
  {
    mContainer[MI::kDelayMask].set(MT::kIndex, DimensionSizes(1, 1, params.getVelocitySourceIndexSize()), kLoad, kNoCheckpoint, kDelayMaskName);
    mContainer[MI::kTransducerSourceInput].set(MT::kReal, DimensionSizes(1, 1, params.getTransducerSourceInputSize()), kLoad, kNoCheckpoint, kTransducerSourceInputName);
  }
  auto seriesDims = [&](size_t many, size_t indexSize, size_t flag) {
    return (many == 0) ? DimensionSizes(1, 1, flag) : DimensionSizes(1, indexSize, flag);
  };
  if (params.getPressureSourceFlag() != 0)
  {
    mContainer[MI::kPressureSourceInput].set(MT::kReal, seriesDims(params.getPressureSourceMany(), params.getPressureSourceIndexSize(), params.getPressureSourceFlag()), kLoad, kNoCheckpoint, kPressureSourceInputName);
    mContainer[MI::kPressureSourceIndex].set(MT::kIndex, DimensionSizes(1, 1, params.getPressureSourceIndexSize()), kLoad, kNoCheckpoint, kPressureSourceIndexName);
  }
  if (params.getVelocityXSourceFlag() != 0)
    mContainer[MI::kVelocityXSourceInput].set(MT::kReal, seriesDims(params.getVelocitySourceMany(), params.getVelocitySourceIndexSize(), params.getVelocityXSourceFlag()), kLoad, kNoCheckpoint, kVelocityXSourceInputName);
  if (params.getVelocityYSourceFlag() != 0)
    mContainer[MI::kVelocityYSourceInput].set(MT::kReal, seriesDims(params.getVelocitySourceMany(), params.getVelocitySourceIndexSize(), params.getVelocityYSourceFlag()), kLoad, kNoCheckpoint, kVelocityYSourceInputName);
  if (params.getVelocityZSourceFlag() != 0)
    mContainer[MI::kVelocityZSourceInput].set(MT::kReal, seriesDims(params.getVelocitySourceMany(), params.getVelocitySourceIndexSize(), params.getVelocityZSourceFlag()), kLoad, kNoCheckpoint, kVelocityZSourceInputName);
  if (((params.getVelocitySourceMode() == Parameters::SourceMode::kAdditive) ||
       (params.getPressureSourceMode() == Parameters::SourceMode::kAdditive)) &&
      (params.getPressureSourceFlag() || params.getVelocityXSourceFlag() || params.getVelocityYSourceFlag() ||
       params.getVelocityZSourceFlag()))
    mContainer[MI::kSourceKappa].set(MT::kReal, reducedDims, kNoLoad, kNoCheckpoint, "source_kappa_r");

  // ---- non-staggered velocity (MatrixContainer.cpp:330-385) ----
  if (params.needsShiftedVelocity())
  {
    const size_t nxR = fullDims.nx / 2 + 1, nyR = fullDims.ny / 2 + 1, nzR = fullDims.nz / 2 + 1;
    const size_t nzRGlobal = nzGlobal / 2 + 1; // z lines (and z_shift_neg_r) keep their global length on a slab
    const size_t xCut = nxR * fullDims.ny * fullDims.nz, yCut = fullDims.nx * nyR * fullDims.nz,
                 zCut = fullDims.nx * fullDims.ny * nzR;
    DimensionSizes shiftDims = fullDims;
    if ((xCut >= yCut) && (xCut >= zCut)) shiftDims.nx = nxR;
    else if ((yCut >= xCut) && (yCut >= zCut)) shiftDims.ny = nyR;
    else shiftDims.nz = nzR;
    mContainer[MI::kTempHipFftShift].set(MT::kFft, shiftDims, kNoLoad, kNoCheckpoint, "hipfft_shift_temp");
    mContainer[MI::kUxShifted].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "ux_shifted");
    mContainer[MI::kUyShifted].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "uy_shifted");
    if (params.isSimulation3D()) mContainer[MI::kUzShifted].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "uz_shifted");
    mContainer[MI::kXShiftNegR].set(MT::kComplex, DimensionSizes(nxR, 1, 1), kLoad, kNoCheckpoint, kXShiftNegRName);
    mContainer[MI::kYShiftNegR].set(MT::kComplex, DimensionSizes(1, nyR, 1), kLoad, kNoCheckpoint, kYShiftNegRName);
    if (params.isSimulation3D())
      mContainer[MI::kZShiftNegR].set(MT::kComplex, DimensionSizes(1, 1, nzRGlobal), kLoad, kNoCheckpoint, kZShiftNegRName);
  }

  // ---- temporaries (MatrixContainer.cpp:387-410): alpha_coeff is loaded *into* Temp1 ----
  if ((params.getAbsorbingFlag() != 0) && (!params.getAlphaCoeffScalarFlag()))
    mContainer[MI::kTemp1RealND].set(MT::kReal, fullDims, kLoad, kNoCheckpoint, kAlphaCoeffName);
  else
    mContainer[MI::kTemp1RealND].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "Temp_1_RS3D");
  mContainer[MI::kTemp2RealND].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "Temp_2_RS3D");
  mContainer[MI::kTemp3RealND].set(MT::kReal, fullDims, kNoLoad, kNoCheckpoint, "Temp_3_RS3D");
  mContainer[MI::kTempHipFftX].set(MT::kFft, reducedDims, kNoLoad, kNoCheckpoint, "hipfft_X_temp");
  mContainer[MI::kTempHipFftY].set(MT::kFft, reducedDims, kNoLoad, kNoCheckpoint, "hipfft_Y_temp");
  mContainer[MI::kTempHipFftZ].set(MT::kFft, reducedDims, kNoLoad, kNoCheckpoint, "hipfft_Z_temp");
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
