// KSpaceFirstOrderSolver.h — orchestrator of the k-space first-order time loop on MI355X.
// Mirror of KSpaceSolver/KSpaceFirstOrderSolver.{h,cpp} for the per-step path and the generators it needs
// (SURVEY.md §8 a1-a9, a12): allocateMemory (:124-141), loadInputData (:159-261), compute (:268-439) =
// initializeFftPlans (:747-777) + preProcessing (:784-857) + computeMainLoop (:864-943) + postProcessing (:950-1053),
// the step pieces (:2087-2396, :2714-2735) and the host generators (:2404-2703).
// Same method names; the device work goes through namespace SolverHipKernels / HipFftComplexMatrix /
// OutputStreamsHipKernels.  Options::fusedKernels selects the MI355X-fused step (fewer, wider kernels, identical
// arithmetic order) instead of one launch per reference kernel.
#ifndef KW_HOST_KSPACE_FIRST_ORDER_SOLVER_H
#define KW_HOST_KSPACE_FIRST_ORDER_SOLVER_H
#include "MatrixContainer.h"
#include "OutputStreams.h"
#include "Parameters.h"

class KSpaceFirstOrderSolver
{
 public:
  using SD = Parameters::SimulationDimension;
  KSpaceFirstOrderSolver();
  virtual ~KSpaceFirstOrderSolver();

  virtual void allocateMemory();
  virtual void freeMemory();
  virtual void loadInputData(const InputProvider& input);
  /// whole simulation: plans, pre-processing, main loop to Nt, post-processing
  virtual void compute();

  // ---- the same pipeline in pieces, so that bench.py / tests can time or inspect the loop alone ----
  void generateInitialDenisty();        // dt/rho0_sg for non-uniform grids (:2650-2685; the reference's spelling)
  void prepare();                       // initializeFftPlans + preProcessing + constants + copyMatricesToDevice
  void runTimeSteps(size_t nSteps);     // body of computeMainLoop for nSteps (stops at Nt)
  void finish();                        // last delayed flush + postProcessing
  void postProcessStoredOutput();       // --post: post-processing of the series an existing output file holds

  MatrixContainer&       getMatrixContainer() { return mMatrixContainer; }
  OutputStreamContainer& getOutputStreamContainer() { return mOutputStreamContainer; }
  /// true once prepare() has chosen the hand-written FFT pipeline for this grid (false: rocFFT + one kernel per stage)
  bool usesFusedPipeline() const { return mFused; }
  /// wall-clock seconds spent so far in: 0 data loading, 1 pre-processing, 2 the time loop, 3 post-processing — what
  /// the reference writes into the header of the output file (Hdf5FileHeader.cpp:372-384)
  double getPhaseTime(int phase) const { return mPhaseTime[phase]; }
  /// "kspaceFirstOrder-HIP" code name (reference: getCodeName, KSpaceFirstOrderSolver.h)
  std::string getCodeName() const { return "kspaceFirstOrder-HIP v0.1 (gfx950)"; }

 protected:
  void initializeFftPlans();
  template<SD simulationDimension> void preProcessing();
  template<SD simulationDimension> void computeMainLoop();
  template<SD simulationDimension> void postProcessing();
  void storeSensorData();

  template<SD simulationDimension> void computeVelocity();
  template<SD simulationDimension> void computeVelocityGradient();
  template<SD simulationDimension> void computeDensityNonliner();
  template<SD simulationDimension> void computeDensityLinear();
  void fusedDensity(bool nonlinear);
  template<SD simulationDimension> void computePressureNonlinear();
  template<SD simulationDimension> void computePressureLinear();
  void addVelocitySource();
  template<SD simulationDimension> void addPressureSource();
  void scaleSource(RealMatrix& scaledSource, const RealMatrix& sourceInput, const IndexMatrix& sourceIndex, const size_t manyFlag);
  template<SD simulationDimension> void addInitialPressureSource();
  template<SD simulationDimension> void computeShiftedVelocity();
  // post-processing of the stored series (KSpaceFirstOrderSolver.cpp:1231-1534, :1783-2080)
  void computeAverageIntensities();
  void computeAverageIntensitiesC();
  void computeQTerm(OutputStreamContainer::OutputStreamIdx intensityX, OutputStreamContainer::OutputStreamIdx intensityY,
                    OutputStreamContainer::OutputStreamIdx intensityZ, OutputStreamContainer::OutputStreamIdx qTerm);
  std::vector<size_t> sensorGridIndices(); // grid index of every sensor point, in stream-buffer order

  void generateKappa();
  void generateSourceKappa();
  void generateKappaAndNablas();
  void generateTauAndEta();
  void computeC2();

  // matrix shortcuts (reference: inline getters at KSpaceFirstOrderSolver.h:560-1070)
  using MI = MatrixContainer::MatrixIdx;
  RealMatrix& real(MI idx) { return mMatrixContainer.getMatrix<RealMatrix>(idx); }
  HipFftComplexMatrix& fft(MI idx) { return mMatrixContainer.getMatrix<HipFftComplexMatrix>(idx); }
  IndexMatrix& index(MI idx) { return mMatrixContainer.getMatrix<IndexMatrix>(idx); }
  RealMatrix& getP() { return real(MI::kP); }
  RealMatrix& getTemp1RealND() { return real(MI::kTemp1RealND); }
  RealMatrix& getTemp2RealND() { return real(MI::kTemp2RealND); }
  RealMatrix& getTemp3RealND() { return real(MI::kTemp3RealND); }
  HipFftComplexMatrix& getTempHipFftX() { return fft(MI::kTempHipFftX); }
  HipFftComplexMatrix& getTempHipFftY() { return fft(MI::kTempHipFftY); }
  HipFftComplexMatrix& getTempHipFftZ() { return fft(MI::kTempHipFftZ); }

 private:
  // ---- MI355X fused spectral pipeline (Options::fusedKernels; csrc/kw_fused.hip) ----
  void   initializeFusedPipeline();
  void   releaseFusedPipeline();
  float* importPadded(MI idx);

  MatrixContainer       mMatrixContainer;
  OutputStreamContainer mOutputStreamContainer;
  Parameters&           mParameters;
  bool                  mPrepared = false;
  double                mPhaseTime[4] = {0.0, 0.0, 0.0, 0.0};
  bool                  mFused    = false;   // fused pipeline active for this grid
  bool                  mOwnsComm = false;   // this solver gave the context its RCCL communicator (slab mode)
  bool                  mTermsFused = false; // pressure terms of this step already produced by the density stage
  bool                  mVelocityChained = false; // x-spectra of u handed over by the velocity stage this step
  bool mPressureFused = false;     // lossless: p of this step already produced by the density stage
  bool mPressureInScratch = false; // the spectrum of p is still in the pipeline scratch (chained by the pressure sum)
  // launch-bound grids: the kernel launches of one steady-state step (no source active, p's spectrum chained from the
  // previous step) are recorded once and replayed (kw_graph_*)
  kw_graph*             mStepGraph = nullptr;
  bool                  mUseStepGraph = false, mStepGraphFailed = false;
  float*                mShiftFilter[3] = {nullptr, nullptr, nullptr}; // kw_fused_shift_velocity filters (x, y, z)
  float*                mKappaPadded = nullptr;
  float*                mNabla1Padded = nullptr;
  float*                mNabla2Padded = nullptr;
  float*                mSourceKappaPadded = nullptr;
};
#endif
