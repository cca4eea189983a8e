// SolverHipKernels.cpp — see SolverHipKernels.h.  Pointer unpacking follows the reference wrappers
// (e.g. SolverCudaKernels.cu:222-243 for the velocity update): the same matrices are fetched from the container and
// a missing per-voxel array (scalar medium) is passed as nullptr, which selects the scalar from the device constants.
#include "SolverHipKernels.h"

#include "HipError.h"

namespace SolverHipKernels
{
using MI = MatrixContainer::MatrixIdx;

static kw_ctx* ctx() { return Parameters::getInstance().getHipParameters().getContext(); }
static float*  real(const MatrixContainer& c, MI idx) { return c.getMatrix<RealMatrix>(idx).getDeviceData(); }
static float*  cplx(const MatrixContainer& c, MI idx) { return c.getMatrix<ComplexMatrix>(idx).getDeviceData(); }

int getHipCodeVersion() { return 950; }

template<SD sd> void computeVelocityHeterogeneous(const MatrixContainer& c)
{
  kwCheck(kw_compute_velocity(ctx(), real(c, MI::kUxSgx), real(c, MI::kUySgy), real(c, MI::kUzSgz),
                              real(c, MI::kTemp1RealND), real(c, MI::kTemp2RealND), real(c, MI::kTemp3RealND),
                              real(c, MI::kDtRho0Sgx), real(c, MI::kDtRho0Sgy), real(c, MI::kDtRho0Sgz),
                              real(c, MI::kPmlXSgx), real(c, MI::kPmlYSgy), real(c, MI::kPmlZSgz)));
}
template<SD sd> void computeVelocityHomogeneousUniform(const MatrixContainer& c)
{
  kwCheck(kw_compute_velocity(ctx(), real(c, MI::kUxSgx), real(c, MI::kUySgy), real(c, MI::kUzSgz),
                              real(c, MI::kTemp1RealND), real(c, MI::kTemp2RealND), real(c, MI::kTemp3RealND), nullptr,
                              nullptr, nullptr, real(c, MI::kPmlXSgx), real(c, MI::kPmlYSgy), real(c, MI::kPmlZSgz)));
}
void addTransducerSource(const MatrixContainer& c)
{
  kwCheck(kw_add_transducer_source(ctx(), real(c, MI::kUxSgx),
                                   (const uint64_t*)c.getMatrix<IndexMatrix>(MI::kVelocitySourceIndex).getDeviceData(),
                                   real(c, MI::kTransducerSourceInput),
                                   (const uint64_t*)c.getMatrix<IndexMatrix>(MI::kDelayMask).getDeviceData(),
                                   Parameters::getInstance().getTimeIndex()));
}
void addVelocitySource(RealMatrix& velocity, const RealMatrix& input, const IndexMatrix& index)
{
  kwCheck(kw_add_velocity_source(ctx(), velocity.getDeviceData(), input.getDeviceData(),
                                 (const uint64_t*)index.getDeviceData(), Parameters::getInstance().getTimeIndex()));
}
template<SD sd> void addPressureSource(const MatrixContainer& c)
{
  kwCheck(kw_add_pressure_source(ctx(), real(c, MI::kRhoX), real(c, MI::kRhoY), real(c, MI::kRhoZ),
                                 real(c, MI::kPressureSourceInput),
                                 (const uint64_t*)c.getMatrix<IndexMatrix>(MI::kPressureSourceIndex).getDeviceData(),
                                 Parameters::getInstance().getTimeIndex()));
}
void insertSourceIntoScalingMatrix(RealMatrix& scaled, const RealMatrix& input, const IndexMatrix& index,
                                   const size_t manyFlag)
{
  kwCheck(kw_insert_source_into_scaling_matrix(ctx(), scaled.getDeviceData(), input.getDeviceData(),
                                               (const uint64_t*)index.getDeviceData(), index.size(), manyFlag != 0,
                                               Parameters::getInstance().getTimeIndex()));
}
void computeSourceGradient(HipFftComplexMatrix& spectrum, const RealMatrix& sourceKappa)
{
  kwCheck(kw_compute_source_gradient(ctx(), spectrum.getDeviceData(), sourceKappa.getDeviceData()));
}
void addVelocityScaledSource(RealMatrix& velocity, const RealMatrix& scaled)
{
  kwCheck(kw_add_velocity_scaled_source(ctx(), velocity.getDeviceData(), scaled.getDeviceData()));
}
template<SD sd> void addPressureScaledSource(const MatrixContainer& c, const RealMatrix& scaled)
{
  kwCheck(kw_add_pressure_scaled_source(ctx(), real(c, MI::kRhoX), real(c, MI::kRhoY), real(c, MI::kRhoZ),
                                        scaled.getDeviceData()));
}
template<SD sd> void addInitialPressureSource(const MatrixContainer& c)
{
  kwCheck(kw_add_initial_pressure_source(ctx(), real(c, MI::kP), real(c, MI::kRhoX), real(c, MI::kRhoY),
                                         real(c, MI::kRhoZ), real(c, MI::kInitialPressureSourceInput),
                                         c.realDeviceOrNull(MI::kC2)));
}
template<SD sd> void computeInitialVelocityHeterogeneous(const MatrixContainer& c)
{
  kwCheck(kw_compute_initial_velocity(ctx(), real(c, MI::kUxSgx), real(c, MI::kUySgy), real(c, MI::kUzSgz),
                                      real(c, MI::kDtRho0Sgx), real(c, MI::kDtRho0Sgy), real(c, MI::kDtRho0Sgz)));
}
template<SD sd> void computeInitialVelocityHomogeneousUniform(const MatrixContainer& c)
{
  kwCheck(kw_compute_initial_velocity(ctx(), real(c, MI::kUxSgx), real(c, MI::kUySgy), real(c, MI::kUzSgz), nullptr,
                                      nullptr, nullptr));
}
template<SD sd> void computePressureGradient(const MatrixContainer& c)
{
  kwCheck(kw_compute_pressure_gradient(ctx(), cplx(c, MI::kTempHipFftX), cplx(c, MI::kTempHipFftY),
                                       cplx(c, MI::kTempHipFftZ), real(c, MI::kKappa), cplx(c, MI::kDdxKShiftPosR),
                                       cplx(c, MI::kDdyKShiftPos), cplx(c, MI::kDdzKShiftPos)));
}
template<SD sd> void computeVelocityGradientShiftNonuniform(const MatrixContainer& c)
{ // SolverCudaKernels.cuh:293; .cu:1303-1320
  kwCheck(kw_compute_velocity_gradient_shift_nonuniform(ctx(), real(c, MI::kDuxdx), real(c, MI::kDuydy), real(c, MI::kDuzdz),
                                                        real(c, MI::kDxudxn), real(c, MI::kDyudyn), real(c, MI::kDzudzn)));
}
template<SD sd> void computeVelocityGradient(const MatrixContainer& c)
{
  kwCheck(kw_compute_velocity_gradient(ctx(), cplx(c, MI::kTempHipFftX), cplx(c, MI::kTempHipFftY),
                                       cplx(c, MI::kTempHipFftZ), real(c, MI::kKappa), cplx(c, MI::kDdxKShiftNegR),
                                       cplx(c, MI::kDdyKShiftNeg), cplx(c, MI::kDdzKShiftNeg)));
}
template<SD sd> void computeDensityNonlinear(const MatrixContainer& c)
{
  kwCheck(kw_compute_density_nonlinear(ctx(), real(c, MI::kRhoX), real(c, MI::kRhoY), real(c, MI::kRhoZ),
                                       real(c, MI::kPmlX), real(c, MI::kPmlY), real(c, MI::kPmlZ), real(c, MI::kDuxdx),
                                       real(c, MI::kDuydy), real(c, MI::kDuzdz), c.realDeviceOrNull(MI::kRho0)));
}
template<SD sd> void computeDensityLinear(const MatrixContainer& c)
{
  kwCheck(kw_compute_density_linear(ctx(), real(c, MI::kRhoX), real(c, MI::kRhoY), real(c, MI::kRhoZ),
                                    real(c, MI::kPmlX), real(c, MI::kPmlY), real(c, MI::kPmlZ), real(c, MI::kDuxdx),
                                    real(c, MI::kDuydy), real(c, MI::kDuzdz), c.realDeviceOrNull(MI::kRho0)));
}
template<SD sd>
void computePressureTermsNonlinear(RealMatrix& densitySum, RealMatrix& nonlinearTerm, RealMatrix& velocityGradientSum,
                                   const MatrixContainer& c)
{
  kwCheck(kw_compute_pressure_terms_nonlinear(ctx(), densitySum.getDeviceData(), nonlinearTerm.getDeviceData(),
                                              velocityGradientSum.getDeviceData(), real(c, MI::kRhoX),
                                              real(c, MI::kRhoY), real(c, MI::kRhoZ), real(c, MI::kDuxdx),
                                              real(c, MI::kDuydy), real(c, MI::kDuzdz), c.realDeviceOrNull(MI::kBOnA),
                                              c.realDeviceOrNull(MI::kRho0)));
}
template<SD sd>
void computePressureTermsLinear(RealMatrix& densitySum, RealMatrix& velocityGradientSum, const MatrixContainer& c)
{
  kwCheck(kw_compute_pressure_terms_linear(ctx(), densitySum.getDeviceData(), velocityGradientSum.getDeviceData(),
                                           real(c, MI::kRhoX), real(c, MI::kRhoY), real(c, MI::kRhoZ),
                                           real(c, MI::kDuxdx), real(c, MI::kDuydy), real(c, MI::kDuzdz),
                                           c.realDeviceOrNull(MI::kRho0)));
}
void computeAbsorbtionTerm(HipFftComplexMatrix& fftPart1, HipFftComplexMatrix& fftPart2, const RealMatrix& absorbNabla1,
                           const RealMatrix& absorbNabla2)
{
  kwCheck(kw_compute_absorbtion_term(ctx(), fftPart1.getDeviceData(), fftPart2.getDeviceData(),
                                     absorbNabla1.getDeviceData(), absorbNabla2.getDeviceData()));
}
void sumPressureTermsNonlinear(const RealMatrix& nonlinearTerm, const RealMatrix& absorbTauTerm,
                               const RealMatrix& absorbEtaTerm, const MatrixContainer& c)
{
  kwCheck(kw_sum_pressure_terms_nonlinear(ctx(), real(c, MI::kP), nonlinearTerm.getDeviceData(),
                                          absorbTauTerm.getDeviceData(), absorbEtaTerm.getDeviceData(),
                                          c.realDeviceOrNull(MI::kC2), c.realDeviceOrNull(MI::kAbsorbTau),
                                          c.realDeviceOrNull(MI::kAbsorbEta)));
}
void sumPressureTermsLinear(const RealMatrix& absorbTauTerm, const RealMatrix& absorbEtaTerm,
                            const RealMatrix& densitySum, const MatrixContainer& c)
{
  kwCheck(kw_sum_pressure_terms_linear(ctx(), real(c, MI::kP), absorbTauTerm.getDeviceData(),
                                       absorbEtaTerm.getDeviceData(), densitySum.getDeviceData(),
                                       c.realDeviceOrNull(MI::kC2), c.realDeviceOrNull(MI::kAbsorbTau),
                                       c.realDeviceOrNull(MI::kAbsorbEta)));
}
template<SD sd> void sumPressureNonlinearLossless(const MatrixContainer& c)
{
  kwCheck(kw_sum_pressure_nonlinear_lossless(ctx(), real(c, MI::kP), real(c, MI::kRhoX), real(c, MI::kRhoY),
                                             real(c, MI::kRhoZ), c.realDeviceOrNull(MI::kC2),
                                             c.realDeviceOrNull(MI::kBOnA), c.realDeviceOrNull(MI::kRho0)));
}
template<SD sd> void sumPressureLinearLossless(const MatrixContainer& c)
{
  kwCheck(kw_sum_pressure_linear_lossless(ctx(), real(c, MI::kP), real(c, MI::kRhoX), real(c, MI::kRhoY),
                                          real(c, MI::kRhoZ), c.realDeviceOrNull(MI::kC2)));
}
void computeVelocityShiftInX(HipFftComplexMatrix& t, const ComplexMatrix& s)
{
  kwCheck(kw_compute_velocity_shift(ctx(), 0, t.getDeviceData(), s.getDeviceData()));
}
void computeVelocityShiftInY(HipFftComplexMatrix& t, const ComplexMatrix& s)
{
  kwCheck(kw_compute_velocity_shift(ctx(), 1, t.getDeviceData(), s.getDeviceData()));
}
void computeVelocityShiftInZ(HipFftComplexMatrix& t, const ComplexMatrix& s)
{
  kwCheck(kw_compute_velocity_shift(ctx(), 2, t.getDeviceData(), s.getDeviceData()));
}

// explicit instances (3-D)
template void computeVelocityHeterogeneous<SD::k3D>(const MatrixContainer&);
template void computeVelocityHomogeneousUniform<SD::k3D>(const MatrixContainer&);
template void addPressureSource<SD::k3D>(const MatrixContainer&);
template void addPressureScaledSource<SD::k3D>(const MatrixContainer&, const RealMatrix&);
template void addInitialPressureSource<SD::k3D>(const MatrixContainer&);
template void computeInitialVelocityHeterogeneous<SD::k3D>(const MatrixContainer&);
template void computeInitialVelocityHomogeneousUniform<SD::k3D>(const MatrixContainer&);
template void computePressureGradient<SD::k3D>(const MatrixContainer&);
template void computeVelocityGradient<SD::k3D>(const MatrixContainer&);
template void computeVelocityGradientShiftNonuniform<SD::k3D>(const MatrixContainer&);
template void computeDensityNonlinear<SD::k3D>(const MatrixContainer&);
template void computeDensityLinear<SD::k3D>(const MatrixContainer&);
template void computePressureTermsNonlinear<SD::k3D>(RealMatrix&, RealMatrix&, RealMatrix&, const MatrixContainer&);
template void computePressureTermsLinear<SD::k3D>(RealMatrix&, RealMatrix&, const MatrixContainer&);
template void sumPressureNonlinearLossless<SD::k3D>(const MatrixContainer&);
template void sumPressureLinearLossless<SD::k3D>(const MatrixContainer&);
} // namespace SolverHipKernels
