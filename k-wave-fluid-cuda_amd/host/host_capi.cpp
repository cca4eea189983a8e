// host_capi.cpp — extern "C" entry points of libkwave_host.so (include/kwave_host.h).
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>
#include <string>

#include "HipError.h"
#include "CompressHelper.h"
#include "HostSolverHandle.h"
#include "MatrixNames.h"
#include "KSpaceFirstOrderSolver.h"
#include "kwave_host.h"

static thread_local std::string g_err;
void kwh_set_error(const std::string& e) { g_err = e; }

Parameters::Options kwh_convert_options(const kwh_options* o)
{
  Parameters::Options opt;
  opt.deviceIdx              = o->device_idx;
  opt.fusedKernels           = o->fused_kernels != 0;
  opt.samplingStartTimeIndex = o->sampling_start_time_index;
  opt.benchmarkTimeStepCount = o->benchmark_time_steps;
  opt.storePressureRaw = o->p_raw; opt.storePressureRms = o->p_rms; opt.storePressureMax = o->p_max;
  opt.storePressureMin = o->p_min; opt.storePressureMaxAll = o->p_max_all; opt.storePressureMinAll = o->p_min_all;
  opt.storePressureFinalAll = o->p_final;
  opt.storeVelocityRaw = o->u_raw; opt.storeVelocityRms = o->u_rms; opt.storeVelocityMax = o->u_max;
  opt.storeVelocityMin = o->u_min; opt.storeVelocityMaxAll = o->u_max_all; opt.storeVelocityMinAll = o->u_min_all;
  opt.storeVelocityFinalAll = o->u_final; opt.storeVelocityNonStaggeredRaw = o->u_non_staggered_raw;
  opt.storePressureC = o->p_c; opt.storeVelocityNonStaggeredC = o->u_non_staggered_c;
  opt.storeIntensityAvgC = o->i_avg_c; opt.noCompressionOverlap = o->no_overlap;
  opt.storeIntensityAvg = o->i_avg; opt.storeQTerm = o->q_term; opt.storeQTermC = o->q_term_c;
  opt.storeVelocityC = o->u_c; opt.frequency = o->frequency;
  opt.onlyPostProcessing = o->only_post_processing != 0;
  opt.complex40bit = o->complex_40bit != 0;
  opt.period = o->period; opt.mos = o->mos ? o->mos : 1; opt.harmonics = o->harmonics ? o->harmonics : 1;
  opt.slabRanks = o->slab_ranks ? o->slab_ranks : 1;
  opt.slabRank  = o->slab_rank;
  opt.nzGlobal  = o->nz_global;
  opt.commUniqueId = o->comm_unique_id;
  opt.exchangeFn   = reinterpret_cast<kw_exchange_fn>(o->exchange_fn);
  opt.exchangeUser = o->exchange_user;
  opt.exchangeStartFn = reinterpret_cast<kw_exchange_start_fn>(o->exchange_start_fn);
  opt.exchangeWaitFn  = reinterpret_cast<kw_exchange_wait_fn>(o->exchange_wait_fn);
  opt.exchangePieceFn = reinterpret_cast<kw_exchange_piece_fn>(o->exchange_piece_fn);
  for (int i = 0; i < 6; i++) opt.scratch[i] = o->scratch[i];
  if (o->tuning != nullptr) { opt.hasTuning = true; opt.tuning = *static_cast<const kw_tuning*>(o->tuning); }
  opt.stepGraph     = o->step_graph != 0;
  opt.commP2P       = o->comm_p2p != 0;
  opt.allgatherFn   = reinterpret_cast<int (*)(void*, const void*, void*, size_t)>(o->comm_allgather_fn);
  opt.allgatherUser = o->comm_allgather_user;
  if (o->rccl_library != nullptr) opt.rcclLibrary = o->rccl_library;
  opt.p2pEmulateLinkGbs   = o->p2p_emulate_link_gbs;
  opt.p2pEmulateLatencyUs = o->p2p_emulate_latency_us;
  return opt;
}

/// Parameters::init + selectDevice + allocateMemory + loadInputData (main.cpp:857-917) on any InputProvider
void kwh_build_solver(kwh_solver& s, const InputProvider& fileInput, const Parameters::Options& opt)
{
  // Nz == 1 selects the 2-D simulation (Parameters.h:175-181 of the reference): complete the z datasets (see adapter)
  size_t nz = 0;
  fileInput.readScalarValue(kNzName, nz);
  const Input2DAdapter adapter(fileInput);
  const InputProvider& input = (nz == 1) ? static_cast<const InputProvider&>(adapter) : fileInput;
  KWH_BIND(&s)
  Parameters& params = Parameters::getInstance();
  params.init(input, opt);
  params.selectDevice();
  params.getHipParameters().setUpDeviceConstants();
  s.solver.reset(new KSpaceFirstOrderSolver());
  s.solver->allocateMemory();
  s.solver->loadInputData(input);
}


#define KWH_TRY try {
#define KWH_CATCH                                                                                                      \
  }                                                                                                                    \
  catch (const std::bad_alloc&) { g_err = "out of memory"; return 4; }                                                 \
  catch (const std::exception& e) { g_err = e.what(); return 1; }                                                      \
  catch (...) { g_err = "unknown exception"; return 1; }                                                               \
  return 0;

extern "C" {

const char* kwh_last_error(void) { return g_err.c_str(); }

int kwh_create(const kwh_dataset* datasets, size_t n, const kwh_options* o, kwh_solver** out)
{
  KWH_TRY
  if (!datasets || !o || !out) throw std::invalid_argument("kwh_create: NULL argument");
  *out = nullptr;
  std::unique_ptr<kwh_solver> s(new kwh_solver());
  for (size_t i = 0; i < n; i++)
  {
    const kwh_dataset& d = datasets[i];
    DimensionSizes dims(d.nx, d.ny, d.nz);
    // complex datasets come with their float count in nx*ny*nz already doubled by the caller's shape
    s->input.add(d.name, d.data, d.dtype == 0 ? InputProvider::DataType::kFloat : InputProvider::DataType::kLong, dims);
  }
  kwh_build_solver(*s, s->input, kwh_convert_options(o));
  *out = s.release();
  KWH_CATCH
}

int kwh_destroy(kwh_solver* s)
{
  KWH_TRY
  if (s)
  {
    {
      KWH_BIND(s)
      if (s->solver) kw_sync(Parameters::getInstance().getHipParameters().getContext());
      s->series_writer.reset(); // drains and closes the streamed output datasets while the streams still exist
      s->solver.reset();
    }
    delete s; // releases the device context with the parameter set
  }
  KWH_CATCH
}

int kwh_run(kwh_solver* s, uint64_t n_steps)
{
  KWH_TRY
  KWH_BIND(s)
  if (!s) throw std::invalid_argument("kwh_run: NULL solver");
  s->solver->runTimeSteps(n_steps);
  KWH_CATCH
}

int kwh_finish(kwh_solver* s)
{
  KWH_TRY
  KWH_BIND(s)
  if (!s) throw std::invalid_argument("kwh_finish: NULL solver");
  s->solver->finish();
  KWH_CATCH
}

int kwh_sync(kwh_solver* s)
{
  KWH_TRY
  KWH_BIND(s)
  (void)s;
  kwCheck(kw_sync(Parameters::getInstance().getHipParameters().getContext()));
  KWH_CATCH
}

uint64_t kwh_time_index(const kwh_solver* s) { return s ? s->params->getTimeIndex() : 0; }
void*    kwh_context(kwh_solver* s) { return s ? s->params->getHipParameters().getContext() : nullptr; }

static BaseMatrix* findMatrix(kwh_solver* s, const char* name, MatrixRecord::MatrixType* type)
{
  for (auto& rec : s->solver->getMatrixContainer().records())
    if (rec.second.matrixName == name)
    {
      *type = rec.second.matrixType;
      return rec.second.matrixPtr;
    }
  throw std::invalid_argument(std::string("no matrix named ") + name + " in this simulation");
}

int kwh_matrix_size(kwh_solver* s, const char* name, uint64_t* n)
{
  KWH_TRY
  KWH_BIND(s)
  MatrixRecord::MatrixType t;
  BaseMatrix* m = findMatrix(s, name, &t);
  if (t == MatrixRecord::MatrixType::kIndex) throw std::invalid_argument("index matrices are not float matrices");
  *n = static_cast<BaseFloatMatrix*>(m)->capacity();
  KWH_CATCH
}

int kwh_get_matrix(kwh_solver* s, const char* name, float* dst, uint64_t n)
{
  KWH_TRY
  KWH_BIND(s)
  MatrixRecord::MatrixType t;
  BaseMatrix* m = findMatrix(s, name, &t);
  if (t == MatrixRecord::MatrixType::kIndex) throw std::invalid_argument("index matrices are not float matrices");
  BaseFloatMatrix* f = static_cast<BaseFloatMatrix*>(m);
  if (f->capacity() != n) throw std::invalid_argument(std::string("size mismatch for matrix ") + name);
  kwCheck(kw_memcpy_d2h(Parameters::getInstance().getHipParameters().getContext(), dst, f->getDeviceData(), n * sizeof(float)));
  KWH_CATCH
}

int kwh_get_scalar(kwh_solver* s, const char* name, float* out)
{
  KWH_TRY
  KWH_BIND(s)
  const Parameters& p = Parameters::getInstance();
  const std::string n(name);
  if (n == "fused_pipeline")
  {
    if (!s) throw std::invalid_argument("kwh_get_scalar: NULL solver");
    *out = s->solver->usesFusedPipeline() ? 1.f : 0.f;
  }
  else if (n == "absorb_tau") *out = p.getAbsorbTauScalar();
  else if (n == "absorb_eta") *out = p.getAbsorbEtaScalar();
  else if (n == "c2") *out = p.getC2Scalar();
  else if (n == "dt_rho0_sgx") *out = p.getDtRho0SgxScalar();
  else if (n == "dt_rho0_sgy") *out = p.getDtRho0SgyScalar();
  else if (n == "dt_rho0_sgz") *out = p.getDtRho0SgzScalar();
  else if (n == "dt") *out = p.getDt();
  else throw std::invalid_argument("unknown scalar " + n);
  KWH_CATCH
}

int kwh_stream_info(kwh_solver* s, const char* name, uint64_t* size, uint64_t* steps)
{
  KWH_TRY
  KWH_BIND(s)
  BaseOutputStream* st = s->solver->getOutputStreamContainer().find(name);
  if (!st) throw std::invalid_argument(std::string("no output stream named ") + name);
  *size  = st->size();
  const bool series = st->reduceOp() == BaseOutputStream::ReduceOperator::kNone ||
                      st->reduceOp() == BaseOutputStream::ReduceOperator::kC;
  *steps = series ? st->sampledSteps() : 1; // raw: sampled steps; compressed: emitted frames
  KWH_CATCH
}

int kwh_stream_read(kwh_solver* s, const char* name, float* dst, uint64_t n)
{
  KWH_TRY
  KWH_BIND(s)
  BaseOutputStream* st = s->solver->getOutputStreamContainer().find(name);
  if (!st) throw std::invalid_argument(std::string("no output stream named ") + name);
  st->loadSeries(); // a series streamed to the output file is read back from there
  if (st->dataset().size() != n) throw std::invalid_argument(std::string("size mismatch for stream ") + name);
  std::memcpy(dst, st->dataset().data(), n * sizeof(float));
  st->releaseSeries();
  KWH_CATCH
}

int kwh_set_matrix(kwh_solver* s, const char* name, const float* src, uint64_t n)
{
  KWH_TRY
  KWH_BIND(s)
  if (!s || !name || !src) throw std::invalid_argument("kwh_set_matrix: NULL argument");
  s->solver->prepare();
  MatrixRecord::MatrixType t;
  BaseMatrix* m = findMatrix(s, name, &t);
  if (t == MatrixRecord::MatrixType::kIndex) throw std::invalid_argument("index matrices are not float matrices");
  BaseFloatMatrix* f = static_cast<BaseFloatMatrix*>(m);
  if (f->capacity() != n) throw std::invalid_argument(std::string("size mismatch for matrix ") + name);
  kwCheck(kw_memcpy_h2d(Parameters::getInstance().getHipParameters().getContext(), f->getDeviceData(), src, n * sizeof(float)));
  KWH_CATCH
}

int kwh_set_time_index(kwh_solver* s, uint64_t t)
{
  KWH_TRY
  KWH_BIND(s)
  if (!s) throw std::invalid_argument("kwh_set_time_index: NULL solver");
  if (t > Parameters::getInstance().getNt()) throw std::invalid_argument("kwh_set_time_index: beyond Nt");
  Parameters::getInstance().setTimeIndex(t);
  KWH_CATCH
}

int kwh_stream_count(kwh_solver* s, uint64_t* n)
{
  KWH_TRY
  KWH_BIND(s)
  *n = s->solver->getOutputStreamContainer().names().size();
  KWH_CATCH
}

int kwh_stream_name(kwh_solver* s, uint64_t i, char* out, uint64_t cap)
{
  KWH_TRY
  KWH_BIND(s)
  const std::vector<std::string> names = s->solver->getOutputStreamContainer().names();
  if (i >= names.size() || names[i].size() + 1 > cap) throw std::invalid_argument("kwh_stream_name: bad index or buffer");
  std::memcpy(out, names[i].c_str(), names[i].size() + 1);
  KWH_CATCH
}

int kwh_stream_count_all(kwh_solver* s, uint64_t* n)
{
  KWH_TRY
  KWH_BIND(s)
  *n = s->solver->getOutputStreamContainer().names(true).size();
  KWH_CATCH
}

int kwh_stream_name_all(kwh_solver* s, uint64_t i, char* out, uint64_t cap)
{
  KWH_TRY
  KWH_BIND(s)
  const std::vector<std::string> names = s->solver->getOutputStreamContainer().names(true);
  if (i >= names.size() || names[i].size() + 1 > cap) throw std::invalid_argument("kwh_stream_name_all: bad index or buffer");
  std::memcpy(out, names[i].c_str(), names[i].size() + 1);
  KWH_CATCH
}

int kwh_stream_checkpoint(kwh_solver* s, const char* name, float* dst, uint64_t cap, uint64_t* n_floats, uint64_t* steps)
{
  KWH_TRY
  KWH_BIND(s)
  BaseOutputStream* st = s->solver->getOutputStreamContainer().find(name);
  if (!st) throw std::invalid_argument(std::string("no output stream named ") + name);
  std::vector<float> state;
  size_t sampled = 0;
  st->checkpointState(state, sampled);
  *n_floats = state.size();
  *steps    = sampled;
  if (dst != nullptr)
  {
    if (cap < state.size()) throw std::invalid_argument("kwh_stream_checkpoint: buffer too small");
    std::memcpy(dst, state.data(), state.size() * sizeof(float));
  }
  KWH_CATCH
}

int kwh_stream_restore(kwh_solver* s, const char* name, const float* src, uint64_t n, uint64_t steps)
{
  KWH_TRY
  KWH_BIND(s)
  s->solver->prepare();
  BaseOutputStream* st = s->solver->getOutputStreamContainer().find(name);
  if (!st) throw std::invalid_argument(std::string("no output stream named ") + name);
  st->restoreState(src, n, steps);
  KWH_CATCH
}

int kwh_find_period(const float* signal, uint64_t length, float* period)
{
  KWH_TRY
  if (!signal || !period) throw std::invalid_argument("kwh_find_period: NULL argument");
  *period = CompressHelper::findPeriod(signal, length);
  KWH_CATCH
}

int kwh_pack_complex_40b(const float* v, uint64_t n, uint8_t* packed, int32_t e)
{
  KWH_TRY
  if (!v || !packed) throw std::invalid_argument("kwh_pack_complex_40b: NULL argument");
  for (uint64_t i = 0; i < n; i++) CompressHelper::convertFloatCTo40b(FloatComplex(v[2 * i], v[2 * i + 1]), packed + 5 * i, e);
  KWH_CATCH
}

int kwh_unpack_complex_40b(const uint8_t* packed, uint64_t n, float* v, int32_t e)
{
  KWH_TRY
  if (!v || !packed) throw std::invalid_argument("kwh_unpack_complex_40b: NULL argument");
  for (uint64_t i = 0; i < n; i++)
  {
    FloatComplex c;
    CompressHelper::convert40bToFloatC(packed + 5 * i, c, e);
    v[2 * i]     = c.real();
    v[2 * i + 1] = c.imag();
  }
  KWH_CATCH
}
}
