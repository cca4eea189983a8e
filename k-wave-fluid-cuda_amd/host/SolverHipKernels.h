// SolverHipKernels.h — host-side mirror of namespace SolverCudaKernels (KSpaceSolver/SolverCudaKernels.cuh:52-500):
// same function names, same argument meaning (whole MatrixContainer or explicit matrices), same error behaviour
// (exception on failure).  Each wrapper unpacks raw device pointers and calls the C-ABI of include/kwave_hip.h —
// this is where a maintainer of the reference would bind libkwave_hip.
#ifndef KW_HOST_SOLVER_HIP_KERNELS_H
#define KW_HOST_SOLVER_HIP_KERNELS_H
#include "MatrixContainer.h"
#include "Parameters.h"

namespace SolverHipKernels
{
using SD = Parameters::SimulationDimension;

/// SolverCudaKernels::getCudaCodeVersion (.cuh:77): here the gfx target of the code objects (950)
int getHipCodeVersion();

template<SD simulationDimension = SD::k3D> void computeVelocityHeterogeneous(const MatrixContainer& container);
template<SD simulationDimension = SD::k3D> void computeVelocityHomogeneousUniform(const MatrixContainer& container);
void addTransducerSource(const MatrixContainer& container);
void addVelocitySource(RealMatrix& velocity, const RealMatrix& velocitySourceInput, const IndexMatrix& velocitySourceIndex);
template<SD simulationDimension = SD::k3D> void addPressureSource(const MatrixContainer& container);
void insertSourceIntoScalingMatrix(RealMatrix& scaledSource, const RealMatrix& sourceInput,
                                   const IndexMatrix& sourceIndex, const size_t manyFlag);
void computeSourceGradient(HipFftComplexMatrix& sourceSpectrum, const RealMatrix& sourceKappa);
void addVelocityScaledSource(RealMatrix& velocity, const RealMatrix& scaledSource);
template<SD simulationDimension = SD::k3D>
void addPressureScaledSource(const MatrixContainer& container, const RealMatrix& scaledSource);
template<SD simulationDimension = SD::k3D> void addInitialPressureSource(const MatrixContainer& container);
template<SD simulationDimension = SD::k3D> void computeInitialVelocityHeterogeneous(const MatrixContainer& container);
template<SD simulationDimension = SD::k3D> void computeInitialVelocityHomogeneousUniform(const MatrixContainer& container);
template<SD simulationDimension = SD::k3D> void computePressureGradient(const MatrixContainer& container);
template<SD simulationDimension = SD::k3D> void computeVelocityGradient(const MatrixContainer& container);
template<SD simulationDimension = SD::k3D> void computeVelocityGradientShiftNonuniform(const MatrixContainer& container); // .cuh:293
template<SD simulationDimension = SD::k3D> void computeDensityNonlinear(const MatrixContainer& container);
template<SD simulationDimension = SD::k3D> void computeDensityLinear(const MatrixContainer& container);
template<SD simulationDimension = SD::k3D>
void computePressureTermsNonlinear(RealMatrix& densitySum, RealMatrix& nonlinearTerm, RealMatrix& velocityGradientSum,
                                   const MatrixContainer& container);
template<SD simulationDimension = SD::k3D>
void computePressureTermsLinear(RealMatrix& densitySum, RealMatrix& velocityGradientSum, const MatrixContainer& container);
void computeAbsorbtionTerm(HipFftComplexMatrix& fftPart1, HipFftComplexMatrix& fftPart2, const RealMatrix& absorbNabla1,
                           const RealMatrix& absorbNabla2);
void sumPressureTermsNonlinear(const RealMatrix& nonlinearTerm, const RealMatrix& absorbTauTerm,
                               const RealMatrix& absorbEtaTerm, const MatrixContainer& container);
void sumPressureTermsLinear(const RealMatrix& absorbTauTerm, const RealMatrix& absorbEtaTerm,
                            const RealMatrix& densitySum, const MatrixContainer& container);
template<SD simulationDimension = SD::k3D> void sumPressureNonlinearLossless(const MatrixContainer& container);
template<SD simulationDimension = SD::k3D> void sumPressureLinearLossless(const MatrixContainer& container);
void computeVelocityShiftInX(HipFftComplexMatrix& fftShiftTemp, const ComplexMatrix& xShiftNegR);
void computeVelocityShiftInY(HipFftComplexMatrix& fftShiftTemp, const ComplexMatrix& yShiftNegR);
void computeVelocityShiftInZ(HipFftComplexMatrix& fftShiftTemp, const ComplexMatrix& zShiftNegR);
} // namespace SolverHipKernels
#endif
