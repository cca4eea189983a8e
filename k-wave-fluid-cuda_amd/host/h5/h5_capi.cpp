// h5_capi.cpp — HDF5 entry points of libkwave_host_h5.so: run the solver from a k-Wave input file and write a k-Wave
// output file (main.cpp:840-966 sequence; dataset shapes IndexOutputStream.cpp:91-117, WholeDomainOutputStream.cpp,
// KSpaceFirstOrderSolver.cpp:952-973,1100-1169), plus a writer for synthetic input files.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include <exception>

#include "../CompressHelper.h"
#include "../HipError.h"
#include "../HostSolverHandle.h"
#include "../MatrixNames.h"
#include "Hdf5File.h"
#include "SeriesWriter.h"

#define KWH_TRY try {
#define KWH_CATCH                                                                                                      \
  }                                                                                                                    \
  catch (const std::bad_alloc&) { kwh_set_error("out of memory"); return 4; }                                          \
  catch (const std::exception& e) { kwh_set_error(e.what()); return 1; }                                               \
  catch (...) { kwh_set_error("unknown exception"); return 1; }                                                        \
  return 0;

static Hdf5SeriesWriter* seriesWriter(kwh_solver* s) { return static_cast<Hdf5SeriesWriter*>(s->series_writer.get()); }

// what a reader needs to expand compression coefficients again (IndexOutputStream.cpp:146-157)
static void writeCompressionAttributes(Hdf5File& out, const std::string& name, const CompressedIndexOutputStream& cs)
{
  const CompressHelper& ch = CompressHelper::getInstance();
  out.writeLongLongAttribute(name, "c_harmonics", static_cast<long long>(ch.getHarmonics()));
  out.writeStringAttribute(name, "c_type", "c");
  out.writeFloatAttribute(name, "c_period", ch.getPeriod());
  out.writeLongLongAttribute(name, "c_mos", static_cast<long long>(ch.getMos()));
  out.writeLongLongAttribute(name, "c_shift", cs.shiftedBasis() ? 1 : 0);
  out.writeFloatAttribute(name, "c_complex_size", cs.is40bit() ? 1.25f : 2.0f);
  out.writeLongLongAttribute(name, "c_max_exp", cs.maxExp());
}

// Per-step output (IndexOutputStream.cpp:87-160 create, :348-372 flushRaw; CuboidOutputStream.cpp:95-140, :656-722): the
// output file exists from the start of the run, every stored series owns its dataset(s) in it and appends a hyperslab
// per sampled step (compression streams: per finished frame) through the writer thread of h5/SeriesWriter.h.
// reopen: continue the output file of a checkpointed run (BaseOutputStream::reopen, e.g. IndexOutputStream.cpp:170-215)
void kwh_open_output(kwh_solver* s, const std::string& path, unsigned compressionLevel, bool reopen)
{
  if (s->series_writer) throw std::runtime_error("the output file is open already");
  const Parameters& params = Parameters::getInstance();
  if (params.getTimeIndex() != 0 && !reopen) throw std::runtime_error("kwh_open_output_file: call it before the first time step");
  auto writer = std::make_shared<Hdf5SeriesWriter>(path, compressionLevel, !reopen);
  OutputStreamContainer& streams = s->solver->getOutputStreamContainer();
  MatrixContainer& mc = s->solver->getMatrixContainer();
  const bool cornersMask = mc.has(MatrixContainer::MatrixIdx::kSensorMaskCorners);
  const size_t steps = (params.getNt() > params.getSamplingStartTimeIndex()) ? params.getNt() - params.getSamplingStartTimeIndex() : 0;
  for (const std::string& name : streams.names())
  {
    BaseOutputStream* st = streams.find(name);
    if (!st->isSeries()) continue;
    auto* cs = dynamic_cast<CompressedIndexOutputStream*>(st);
    // rows of the dataset: every sampled step, or every finished frame (IndexOutputStream.cpp:108-117)
    const size_t rows = cs ? std::max<size_t>(steps / CompressHelper::getInstance().getOSize(), 1) : steps;
    std::vector<Hdf5SeriesWriter::Part> parts;
    if (cornersMask && cs == nullptr)
    {
      const IndexMatrix& corners = mc.getMatrix<IndexMatrix>(MatrixContainer::MatrixIdx::kSensorMaskCorners);
      writer->drain();
      if (!reopen) writer->file().createGroup(name);
      size_t offset = 0;
      for (size_t c = 0; c < corners.getDimensionSizes().ny; c++)
      {
        const DimensionSizes a = corners.getTopLeftCorner(c), b = corners.getBottomRightCorner(c);
        Hdf5SeriesWriter::Part p;
        p.name   = name + "/" + std::to_string(c + 1);
        p.dims   = DimensionSizes(b.nx - a.nx + 1, b.ny - a.ny + 1, b.nz - a.nz + 1, rows);
        p.offset = offset;
        p.n      = corners.getSizeOfCuboid(c);
        offset += p.n;
        parts.push_back(p);
      }
    }
    else
    {
      Hdf5SeriesWriter::Part p;
      p.name = name;
      p.dims = DimensionSizes(st->size(), rows, 1);
      p.n    = st->size();
      parts.push_back(p);
    }
    if (rows == 0) continue; // sampling never starts: nothing to store
    st->attachSink(writer->makeSink(parts, st->size(), rows));
    if (cs && !reopen) writeCompressionAttributes(writer->file(), name, *cs);
  }
  s->series_writer = writer;
}

// Header and scalars of the output file (writeOutputDataInfo, KSpaceFirstOrderSolver.cpp:1161-1168 + Parameters.cpp:559-647).
// The reference writes them at the end of EVERY leg of a checkpointed run (compute(), :429-430) — a restart checks the
// file through them (checkOutputFile, :2843-2891) — so this runs at every checkpoint as well as at the end.
static void writeOutputInfo(kwh_solver* s, Hdf5File& out)
{
  const Parameters& params = Parameters::getInstance();
  out.writeHeader("output", "k-Wave output written by kspaceFirstOrder-HIP");
  { // the rest of the output header (Hdf5FileHeader.cpp:78-87, :340-384): host, cores, memory, phase times as strings
    char host[256] = "unknown";
    gethostname(host, sizeof(host) - 1);
    out.writeStringAttribute("/", "host_names", host);
    out.writeStringAttribute("/", "number_of_cpu_cores", std::to_string(std::max(1u, std::thread::hardware_concurrency())));
    long rssKb = 0, peakKb = 0;
    if (FILE* f = std::fopen("/proc/self/status", "r"))
    {
      char line[256];
      while (std::fgets(line, sizeof(line), f))
      {
        std::sscanf(line, "VmRSS: %ld kB", &rssKb);
        std::sscanf(line, "VmHWM: %ld kB", &peakKb);
      }
      std::fclose(f);
    }
    out.writeStringAttribute("/", "total_memory_in_use", std::to_string(rssKb >> 10) + " MB");
    out.writeStringAttribute("/", "peak_core_memory_in_use", std::to_string(peakKb >> 10) + " MB");
    auto seconds = [](double t) { char b[32]; std::snprintf(b, sizeof(b), "%8.2fs", t); return std::string(b); };
    const KSpaceFirstOrderSolver& sv = *s->solver;
    const double total = sv.getPhaseTime(0) + sv.getPhaseTime(1) + sv.getPhaseTime(2) + sv.getPhaseTime(3);
    out.writeStringAttribute("/", "total_execution_time", seconds(total));
    out.writeStringAttribute("/", "data_loading_phase_execution_time", seconds(sv.getPhaseTime(0)));
    out.writeStringAttribute("/", "pre-processing_phase_execution_time", seconds(sv.getPhaseTime(1)));
    out.writeStringAttribute("/", "simulation_phase_execution_time", seconds(sv.getPhaseTime(2)));
    out.writeStringAttribute("/", "post-processing_phase_execution_time", seconds(sv.getPhaseTime(3)));
  }
  const DimensionSizes dims = params.getGlobalDimensionSizes();
  out.writeScalarValue(kNxName, dims.nx);
  out.writeScalarValue(kNyName, dims.ny);
  out.writeScalarValue(kNzName, dims.nz);
  out.writeScalarValue(kNtName, params.getNt());
  out.writeScalarValue(kTimeIndexName, params.getTimeIndex());
  out.writeScalarValue(kDtName, params.getDt());
  out.writeScalarValue(kDxName, params.getDx());
  out.writeScalarValue(kDyName, params.getDy());
  if (params.isSimulation3D()) out.writeScalarValue(kDzName, params.getDz());
  out.writeScalarValue(kCRefName, params.getCRef());
  // the rest of the reference's output scalars (Parameters.cpp:580-647)
  const char* const pmlSizeNames[3]  = {"pml_x_size", "pml_y_size", "pml_z_size"};
  const char* const pmlAlphaNames[3] = {"pml_x_alpha", "pml_y_alpha", "pml_z_alpha"};
  const int axes = params.isSimulation3D() ? 3 : 2;
  for (int a = 0; a < axes; a++) out.writeScalarValue(pmlSizeNames[a], params.getPmlSize(a));
  for (int a = 0; a < axes; a++) out.writeScalarValue(pmlAlphaNames[a], params.getPmlAlpha(a));
  out.writeScalarValue(kPressureSourceFlagName, params.getPressureSourceFlag());
  out.writeScalarValue(kInitialPressureSourceFlagName, params.getInitialPressureSourceFlag());
  out.writeScalarValue("transducer_source_flag", params.getTransducerSourceFlag());
  out.writeScalarValue("ux_source_flag", params.getVelocityXSourceFlag());
  out.writeScalarValue("uy_source_flag", params.getVelocityYSourceFlag());
  if (params.isSimulation3D()) out.writeScalarValue("uz_source_flag", params.getVelocityZSourceFlag());
  out.writeScalarValue(kNonUniformGridFlagName, params.getNonUniformGridFlag());
  out.writeScalarValue(kAbsorbingFlagName, params.getAbsorbingFlag());
  out.writeScalarValue(kNonLinearFlagName, params.getNonLinearFlag());
  if (params.getVelocityXSourceFlag() > 0 || params.getVelocityYSourceFlag() > 0 || params.getVelocityZSourceFlag() > 0)
  {
    out.writeScalarValue(kVelocitySourceManyName, params.getVelocitySourceMany());
    out.writeScalarValue(kVelocitySourceModeName, static_cast<size_t>(params.getVelocitySourceMode()));
  }
  if (params.getPressureSourceFlag() != 0)
  {
    out.writeScalarValue(kPressureSourceManyName, params.getPressureSourceMany());
    out.writeScalarValue(kPressureSourceModeName, static_cast<size_t>(params.getPressureSourceMode()));
  }
  if (params.getAbsorbingFlag() != 0) out.writeScalarValue(kAlphaPowerName, params.getAlphaPower());
  out.writeScalarValue(kSensorMaskTypeName, static_cast<size_t>(params.getSensorMaskType()));
}

// One stream's data as its dataset(s) of the output file: raw series (Nsens, steps, 1); aggregates (Nsens, 1, 1);
// whole-domain (Nx, Ny, Nz); with a corners mask a group per stream and a dataset per cuboid — (nx, ny, nz, steps) for
// series, (nx, ny, nz) for aggregates (CuboidOutputStream.cpp:95-140, :656-722; `data` holds the cuboids back to back per
// step).  Datasets that exist already (an aggregate flushed by an earlier checkpoint) are rewritten in place.
static void writeStreamData(kwh_solver* s, Hdf5File& out, BaseOutputStream* st, const std::string& name, const float* data,
                            bool series, size_t steps)
{
  const DimensionSizes dims = Parameters::getInstance().getGlobalDimensionSizes();
  MatrixContainer& mcs = s->solver->getMatrixContainer();
  if (mcs.has(MatrixContainer::MatrixIdx::kSensorMaskCorners) && dynamic_cast<WholeDomainOutputStream*>(st) == nullptr)
  {
    const IndexMatrix& corners = mcs.getMatrix<IndexMatrix>(MatrixContainer::MatrixIdx::kSensorMaskCorners);
    if (!out.datasetExists(name)) out.createGroup(name);
    size_t offset = 0;
    std::vector<float> block;
    for (size_t c = 0; c < corners.getDimensionSizes().ny; c++)
    {
      const DimensionSizes a = corners.getTopLeftCorner(c), b = corners.getBottomRightCorner(c);
      DimensionSizes cd(b.nx - a.nx + 1, b.ny - a.ny + 1, b.nz - a.nz + 1, series ? steps : 0);
      const size_t n = corners.getSizeOfCuboid(c);
      block.resize(n * steps);
      for (size_t t = 0; t < steps; t++) std::copy_n(data + t * st->size() + offset, n, block.data() + t * n);
      out.writeCuboid(name + "/" + std::to_string(c + 1), cd, block.data());
      offset += n;
    }
    return;
  }
  DimensionSizes d(st->size(), steps, 1);
  if (!series && st->size() == dims.nElements()) d = dims;
  out.writeMatrix(name, d, data, Hdf5File::MatrixDomainType::kReal);
  if (auto* cs = dynamic_cast<CompressedIndexOutputStream*>(st)) writeCompressionAttributes(out, name, *cs);
}
// the inverse for an aggregate: the accumulator a checkpoint left in the output file (reopen(), e.g.
// IndexOutputStream.cpp:213-232, WholeDomainOutputStream.cpp:124-133)
static void readAggregate(kwh_solver* s, Hdf5File& out, BaseOutputStream* st, const std::string& name, std::vector<float>& data)
{
  data.resize(st->size());
  MatrixContainer& mcs = s->solver->getMatrixContainer();
  if (mcs.has(MatrixContainer::MatrixIdx::kSensorMaskCorners) && dynamic_cast<WholeDomainOutputStream*>(st) == nullptr)
  {
    const IndexMatrix& corners = mcs.getMatrix<IndexMatrix>(MatrixContainer::MatrixIdx::kSensorMaskCorners);
    size_t offset = 0;
    for (size_t c = 0; c < corners.getDimensionSizes().ny; c++)
    {
      const size_t n = corners.getSizeOfCuboid(c);
      out.readCompleteDataset(name + "/" + std::to_string(c + 1), n, data.data() + offset);
      offset += n;
    }
    return;
  }
  out.readCompleteDataset(name, data.size(), data.data());
}

void kwh_write_output(kwh_solver* s, const std::string& path, unsigned compressionLevel, bool copySensorMask)
{
  const Parameters& params = Parameters::getInstance();
  Hdf5File fresh;
  Hdf5SeriesWriter* writer = seriesWriter(s);
  if (writer != nullptr)
  { // the file has been open since the start of the run and holds the series already: complete it
    if (writer->path() != path) throw std::invalid_argument("the output is being streamed to " + writer->path() + ", it cannot be written to " + path);
    writer->finish();
    compressionLevel = writer->compressionLevel();
  }
  else
  {
    fresh.create(path);
    fresh.setOutputLayout(true, compressionLevel); // chunked like the reference's output; -c N deflate level
  }
  Hdf5File& out = writer ? writer->file() : fresh;
  writeOutputInfo(s, out);
  const DimensionSizes dims = params.getGlobalDimensionSizes();
  OutputStreamContainer& streams = s->solver->getOutputStreamContainer();
  for (const std::string& name : streams.names())
  {
    BaseOutputStream* st = streams.find(name);
    const bool series = st->reduceOp() == BaseOutputStream::ReduceOperator::kNone ||
                        st->reduceOp() == BaseOutputStream::ReduceOperator::kC;
    if (series && st->hasSink()) continue; // streamed: in the file since the step it was sampled at
    const size_t steps = series ? st->sampledSteps() : 1;
    if (st->dataset().size() != st->size() * steps) continue; // nothing sampled yet
    writeStreamData(s, out, st, name, st->dataset().data(), series, steps);
  }
  auto writeFinal = [&](MatrixContainer::MatrixIdx idx, const std::string& name) {
    RealMatrix& m = s->solver->getMatrixContainer().getMatrix<RealMatrix>(idx);
    m.copyFromDevice();
    out.writeMatrix(name, m.getDimensionSizes(), m.getHostData(), Hdf5File::MatrixDomainType::kReal);
    m.freeHostData();
  };
  using MI = MatrixContainer::MatrixIdx;
  if (params.getStorePressureFinalAllFlag()) writeFinal(MI::kP, kPressureFinalName);
  if (params.getStoreVelocityFinalAllFlag())
  {
    writeFinal(MI::kUxSgx, kUxFinalName);
    writeFinal(MI::kUySgy, kUyFinalName);
    if (params.isSimulation3D()) writeFinal(MI::kUzSgz, kUzFinalName);
  }
  if (copySensorMask)
  { // --copy_sensor_mask (KSpaceFirstOrderSolver.cpp:1036-1052): the mask goes to the output file, 1-based again
    using MI = MatrixContainer::MatrixIdx;
    MatrixContainer& mc = s->solver->getMatrixContainer();
    const MI   idx  = (params.getSensorMaskType() == Parameters::SensorMaskType::kIndex) ? MI::kSensorMaskIndex : MI::kSensorMaskCorners;
    const char* name = (idx == MI::kSensorMaskIndex) ? "sensor_mask_index" : "sensor_mask_corners";
    if (mc.has(idx))
    {
      IndexMatrix& m = mc.getMatrix<IndexMatrix>(idx);
      std::vector<size_t> oneBased(m.getHostData(), m.getHostData() + m.size());
      for (size_t& v : oneBased) v += 1;
      out.writeMatrix(name, m.getDimensionSizes(), oneBased.data());
    }
  }
  out.close();
  s->series_writer.reset();
}
void kwh_write_output(kwh_solver* s, const std::string& path) { kwh_write_output(s, path, 0, false); }

// --post (KSpaceFirstOrderSolver.cpp:373-415, :977-1024): no simulation; the time-averaged intensities and the Q terms are
// computed from the p / u_non_staggered series (I_avg, Q_term) or the coefficient frames (I_avg_c, Q_term_c) that an
// earlier run left in the output file, and are added to that file (replacing older results of the same name).
void kwh_post_process_output(kwh_solver* s, const std::string& path)
{
  const Parameters& params = Parameters::getInstance();
  if (!params.getOnlyPostProcessingFlag()) throw std::invalid_argument("post-processing of an output file needs the --post option set at creation");
  if (!(params.getStoreIntensityAvgFlag() || params.getStoreIntensityAvgCFlag() || params.getStoreQTermFlag() || params.getStoreQTermCFlag()))
    throw std::invalid_argument("--post needs at least one of --I_avg, --I_avg_c, --Q_term, --Q_term_c"); // CommandLineParameters.cpp:931-936
  kwh_open_output(s, path, 0, true);
  Hdf5SeriesWriter* writer = seriesWriter(s);
  OutputStreamContainer& streams = s->solver->getOutputStreamContainer();
  const size_t steps = (params.getNt() > params.getSamplingStartTimeIndex()) ? params.getNt() - params.getSamplingStartTimeIndex() : 0;
  for (const std::string& name : streams.names())
  {
    BaseOutputStream* st = streams.find(name);
    if (!st->isSeries() || !st->hasSink()) continue;
    const bool frames = dynamic_cast<CompressedIndexOutputStream*>(st) != nullptr;
    st->adoptStoredSeries(frames ? std::max<size_t>(steps / CompressHelper::getInstance().getOSize(), 1) : steps);
  }
  s->solver->postProcessStoredOutput();
  writer->finish();
  Hdf5File& out = writer->file();
  using RO = BaseOutputStream::ReduceOperator;
  for (const std::string& name : streams.names())
  {
    BaseOutputStream* st = streams.find(name);
    const RO op = st->reduceOp();
    if (op != RO::kIAvg && op != RO::kIAvgC && op != RO::kQTerm && op != RO::kQTermC) continue;
    if (st->dataset().size() != st->size()) continue;
    if (out.datasetExists(name)) out.remove(name);
    out.writeMatrix(name, DimensionSizes(st->size(), 1, 1), st->dataset().data(), Hdf5File::MatrixDomainType::kReal);
  }
  out.close();
  s->series_writer.reset();
}

// ---- checkpoint file (KSpaceFirstOrderSolver.cpp:1176-1224 write, :186-228 + :1124-1169 read / check) --------------
// Root datasets: the seven state arrays under their matrix names (MatrixContainer.cpp:504-537), t_index, Nx, Ny, Nz;
// header file_type = "checkpoint".
//
// Stream state, the reference's layout — used whenever the output file is open (the command-line program always;
// kwh_open_output_file), so that either code can resume the other's run:
//   * raw series and compression frames are in the output file already (appended step by step);
//   * aggregated streams (rms / max / min, index, cuboid or whole-domain; I_avg_c) are flushed, as the accumulators
//     they are, into their own datasets of the OUTPUT file and read back from there on restart (checkpoint() / reopen():
//     IndexOutputStream.cpp:536-557, 213-232; WholeDomainOutputStream.cpp:233-241, 124-133) — the final values
//     replace them when the run completes;
//   * compression accumulators go to the CHECKPOINT file as Temp_<name>_1 / Temp_<name>_2 (c1, c2) and the running
//     I_avg_c sum as Temp_<name> (BaseOutputStream.cpp:551-606, 520-543);
//   * the number of sampled steps is not stored: it follows from t_index and the sampling start (reopen(), :197-209);
//   * header and scalars of the output file are written at every checkpoint (compute(), :429-430).
// Without an output file (the file-less host API) the streams' state has nowhere to go but the checkpoint file:
// "stream_<name>" (series so far, accumulators) + "stream_<name>_steps" — this build's private layout, also still read.
static const char* const kCheckpointMatrices[] = { "p", "rhox", "rhoy", "rhoz", "ux_sgx", "uy_sgy", "uz_sgz" };
using RO = BaseOutputStream::ReduceOperator;
static bool isAggregate(RO op) { return op == RO::kRms || op == RO::kMax || op == RO::kMin || op == RO::kIAvgC; }

void kwh_checkpoint_write_impl(kwh_solver* s, const std::string& path)
{
  const Parameters& params = Parameters::getInstance();
  s->solver->prepare();
  // written beside the target and renamed over it once complete: a run killed while checkpointing (the situation
  // --checkpoint_interval exists for) keeps its previous restart point instead of a truncated file
  const std::string partial = path + ".partial";
  Hdf5File f;
  f.create(partial);
  f.writeHeader("checkpoint", "k-Wave checkpoint written by kspaceFirstOrder-HIP");
  const DimensionSizes dims = params.getFullDimensionSizes();
  f.writeScalarValue(kTimeIndexName, params.getTimeIndex());
  f.writeScalarValue(kNxName, dims.nx);
  f.writeScalarValue(kNyName, dims.ny);
  f.writeScalarValue(kNzName, dims.nz);
  std::vector<float> buf(dims.nElements());
  for (const char* name : kCheckpointMatrices)
  {
    if (kwh_get_matrix(s, name, buf.data(), buf.size()) != 0) throw std::runtime_error(kwh_last_error());
    f.writeMatrix(name, dims, buf.data(), Hdf5File::MatrixDomainType::kReal);
  }
  OutputStreamContainer& streams = s->solver->getOutputStreamContainer();
  Hdf5SeriesWriter* writer = seriesWriter(s);
  const bool sampled = params.getTimeIndex() > params.getSamplingStartTimeIndex(); // "we're here one step ahead" (:1212-1219)
  for (const std::string& name : streams.names(true))
  {
    BaseOutputStream* st = streams.find(name);
    if (writer == nullptr)
    { // private layout
      std::vector<float> state;
      size_t steps = 0;
      st->checkpointState(state, steps);
      f.writeScalarValue("stream_" + name + "_steps", steps);
      if (!state.empty())
        f.writeMatrix("stream_" + name, DimensionSizes(state.size(), 1, 1), state.data(), Hdf5File::MatrixDomainType::kReal);
      continue;
    }
    const RO op = st->reduceOp();
    if (st->isSeries() && !st->hasSink() && op == RO::kNone && sampled)
      throw std::runtime_error("checkpoint: the series of stream " + name + " is kept in memory only (it feeds a post-processed "
                               "quantity and is not part of the output); such a run cannot be checkpointed in the reference's layout");
    if (!sampled) continue;
    if (auto* cs = dynamic_cast<CompressedIndexOutputStream*>(st))
    {
      std::vector<float> c1, c2;
      cs->accumulators(c1, c2);
      f.writeMatrix("Temp_" + name + "_1", DimensionSizes(c1.size(), 1, 1), c1.data(), Hdf5File::MatrixDomainType::kReal);
      f.writeMatrix("Temp_" + name + "_2", DimensionSizes(c2.size(), 1, 1), c2.data(), Hdf5File::MatrixDomainType::kReal);
      continue;
    }
    std::vector<float> state;
    size_t steps = 0;
    st->checkpointState(state, steps); // raw: flushes what is still in flight; aggregates: the accumulator as it stands
    if (!isAggregate(op)) continue;
    if (op == RO::kIAvgC) f.writeMatrix("Temp_" + name, DimensionSizes(state.size(), 1, 1), state.data(), Hdf5File::MatrixDomainType::kReal);
    if (st->doNotSave()) continue;
    writer->drain();
    writeStreamData(s, writer->file(), st, name, state.data(), false, 1);
  }
  if (writer != nullptr)
  {
    writer->drain();
    writeOutputInfo(s, writer->file());
    H5Fflush(writer->file().handle(), H5F_SCOPE_GLOBAL);
  }
  f.close();
  if (std::rename(partial.c_str(), path.c_str()) != 0)
    throw std::runtime_error("cannot move the finished checkpoint " + partial + " to " + path);
}

void kwh_checkpoint_read_impl(kwh_solver* s, const std::string& path)
{
  const Parameters& params = Parameters::getInstance();
  s->solver->prepare();
  Hdf5File f;
  f.open(path);
  if (f.readFileType() != "checkpoint") throw std::invalid_argument(path + " is not a checkpoint file"); // :1132-1136
  const DimensionSizes dims = params.getFullDimensionSizes();
  size_t v[4] = {0, 0, 0, 0};
  f.readCompleteDataset(kNxName, 1, &v[0]);
  f.readCompleteDataset(kNyName, 1, &v[1]);
  f.readCompleteDataset(kNzName, 1, &v[2]);
  f.readCompleteDataset(kTimeIndexName, 1, &v[3]);
  if (v[0] != dims.nx || v[1] != dims.ny || v[2] != dims.nz)
    throw std::invalid_argument("The dimension sizes in the checkpoint file do not match the input file"); // :1160-1168
  std::vector<float> buf(dims.nElements());
  for (const char* name : kCheckpointMatrices)
  {
    f.readCompleteDataset(name, buf.size(), buf.data());
    if (kwh_set_matrix(s, name, buf.data(), buf.size()) != 0) throw std::runtime_error(kwh_last_error());
  }
  OutputStreamContainer& streams = s->solver->getOutputStreamContainer();
  const std::vector<std::string> names = streams.names(true);
  bool privateLayout = false;
  for (const std::string& name : names) privateLayout = privateLayout || f.datasetExists("stream_" + name + "_steps");
  if (privateLayout)
  {
    for (const std::string& name : names)
    {
      size_t steps = 0;
      if (!f.datasetExists("stream_" + name + "_steps"))
        throw std::invalid_argument(path + " holds no state for the output stream \"" + name + "\"");
      f.readCompleteDataset("stream_" + name + "_steps", 1, &steps);
      std::vector<float> state;
      if (f.datasetExists("stream_" + name))
      {
        state.resize(f.getDatasetSize("stream_" + name));
        f.readCompleteDataset("stream_" + name, state.size(), state.data());
      }
      streams.find(name)->restoreState(state.data(), state.size(), steps);
    }
  }
  else if (!names.empty())
  { // the reference's layout: stream state lives in the re-opened output file and in the Temp_ datasets
    Hdf5SeriesWriter* writer = seriesWriter(s);
    if (writer == nullptr)
      throw std::invalid_argument(path + " is a checkpoint in the reference's layout: the state of the output streams is in the "
                                  "output file of that run — re-open it first (kwh_open_output_file with reopen = 1)");
    writer->drain();
    Hdf5File& out = writer->file();
    if (out.readFileType() != "output") throw std::invalid_argument(writer->path() + " is not the output file of a checkpointed run"); // :2843-2891
    size_t o[3] = {0, 0, 0};
    out.readCompleteDataset(kNxName, 1, &o[0]);
    out.readCompleteDataset(kNyName, 1, &o[1]);
    out.readCompleteDataset(kNzName, 1, &o[2]);
    const DimensionSizes g = params.getGlobalDimensionSizes();
    if (o[0] != g.nx || o[1] != g.ny || o[2] != g.nz)
      throw std::invalid_argument("The dimension sizes in the output file do not match the input file");
    const size_t start = params.getSamplingStartTimeIndex();
    const size_t steps = (v[3] > start) ? v[3] - start : 0; // IndexOutputStream.cpp:197-203
    for (const std::string& name : names)
    {
      BaseOutputStream* st = streams.find(name);
      const RO op = st->reduceOp();
      if (auto* cs = dynamic_cast<CompressedIndexOutputStream*>(st))
      {
        std::vector<float> c1(st->size(), 0.0f), c2(st->size(), 0.0f);
        if (steps > 0)
        {
          f.readCompleteDataset("Temp_" + name + "_1", c1.size(), c1.data());
          f.readCompleteDataset("Temp_" + name + "_2", c2.size(), c2.data());
        }
        cs->restoreAccumulators(c1.data(), c2.data(), c1.size(), steps);
      }
      else if (op == RO::kNone)
      {
        if (!st->hasSink() && steps > 0)
          throw std::invalid_argument("the series of stream " + name + " is not in the output file: this run cannot be resumed from a "
                                      "checkpoint in the reference's layout");
        st->restoreState(nullptr, 0, steps);
      }
      else if (isAggregate(op) && steps > 0)
      {
        std::vector<float> state;
        if (op == RO::kIAvgC)
        { // the running sum, and the number of frames added to it (IndexOutputStream.cpp:204-209)
          state.resize(st->size());
          f.readCompleteDataset("Temp_" + name, state.size(), state.data());
          st->restoreState(state.data(), state.size(), steps / CompressHelper::getInstance().getOSize());
        }
        else
        {
          readAggregate(s, out, st, name, state);
          st->restoreState(state.data(), state.size(), steps);
        }
      }
    }
  }
  if (kwh_set_time_index(s, v[3]) != 0) throw std::runtime_error(kwh_last_error());
  f.close();
}

extern "C" {

KWH_API int kwh_create_from_file(const char* input_path, const kwh_options* o, kwh_solver** out)
{
  KWH_TRY
  if (!input_path || !o || !out) throw std::invalid_argument("kwh_create_from_file: NULL argument");
  *out = nullptr;
  std::unique_ptr<kwh_solver> s(new kwh_solver());
  s->file_input.reset(new Hdf5Input(input_path));
  kwh_build_solver(*s, *s->file_input, kwh_convert_options(o));
  s->file_input.reset(); // everything has been loaded; the file is closed like in loadInputData (:247-252)
  *out = s.release();
  KWH_CATCH
}

KWH_API int kwh_write_output_file(kwh_solver* s, const char* path)
{
  KWH_TRY
  KWH_BIND(s)
  if (!s || !path) throw std::invalid_argument("kwh_write_output_file: NULL argument");
  kwh_write_output(s, path);
  KWH_CATCH
}

/* write an input file from in-memory datasets (the synthetic generator's output); complex[i] != 0 marks interleaved
 * complex datasets (domain_type "complex") */
KWH_API int kwh_write_file(const char* path, const char* file_type, const char* description, const kwh_dataset* sets,
                           size_t n, const int32_t* is_complex);
KWH_API int kwh_write_input_file(const char* path, const kwh_dataset* sets, size_t n, const int32_t* is_complex)
{
  return kwh_write_file(path, "input", "synthetic k-Wave input written by kwave_amd.synthetic", sets, n, is_complex);
}
/* the same for any file type ("input" / "output"): the output file of a slab-decomposed run is assembled by rank 0 from
 * gathered datasets; output files get the chunked layout of the reference's output */
KWH_API int kwh_write_file(const char* path, const char* file_type, const char* description, const kwh_dataset* sets,
                           size_t n, const int32_t* is_complex)
{
  KWH_TRY
  Hdf5File f;
  f.create(path);
  if (std::string(file_type) == "output") f.setOutputLayout(true, 0);
  f.writeHeader(file_type, description);
  for (size_t i = 0; i < n; i++)
  {
    const DimensionSizes d(sets[i].nx, sets[i].ny, sets[i].nz);
    if (sets[i].dtype == 0)
      f.writeMatrix(sets[i].name, d, static_cast<const float*>(sets[i].data),
                    (is_complex && is_complex[i]) ? Hdf5File::MatrixDomainType::kComplex : Hdf5File::MatrixDomainType::kReal);
    else
      f.writeMatrix(sets[i].name, d, static_cast<const size_t*>(sets[i].data));
  }
  f.close();
  KWH_CATCH
}

/* add one cuboid of a corner-mask stream to an existing output file: dataset "<group>/<index>" with dims (nx, ny, nz, nt)
 * — nt = 0 for an aggregate — in the layout of kwh_write_output_file (CuboidOutputStream.cpp:95-140); the group is
 * created with its first cuboid.  Used by the multi-GPU runner, whose rank 0 assembles the cuboids from the slabs. */
KWH_API int kwh_h5_append_cuboid(const char* path, const char* group, uint64_t index, const uint64_t dims[4], const float* data)
{
  KWH_TRY
  if (!path || !group || !dims || !data || index == 0) throw std::invalid_argument("kwh_h5_append_cuboid: invalid argument");
  Hdf5File f;
  f.open(path, false);
  f.setOutputLayout(true, 0);
  if (!f.datasetExists(group)) f.createGroup(group);
  f.writeCuboid(std::string(group) + "/" + std::to_string(index), DimensionSizes(dims[0], dims[1], dims[2], dims[3]), data);
  f.close();
  KWH_CATCH
}

KWH_API int kwh_h5_dataset_exists(const char* path, const char* name, int32_t* exists)
{
  KWH_TRY
  Hdf5File f;
  f.open(path, true);
  *exists = f.datasetExists(name) ? 1 : 0;
  KWH_CATCH
}
KWH_API int kwh_h5_read_planes(const char* path, const char* name, uint64_t z0, uint64_t n_planes, float* dst)
{
  KWH_TRY
  Hdf5File f;
  f.open(path, true);
  f.readPlanes(name, z0, n_planes, dst);
  KWH_CATCH
}

/* generic readers used by the tests: dims as (x,y,z); dtype 0 float / 1 uint64 */
KWH_API int kwh_h5_dataset_info(const char* path, const char* name, uint64_t dims[3], int32_t* dtype, int32_t* is_complex)
{
  KWH_TRY
  Hdf5File f;
  f.open(path, true);
  const DimensionSizes d = f.getDatasetDimensionSizes(name);
  dims[0] = d.nx; dims[1] = d.ny; dims[2] = d.nz;
  *dtype      = f.readMatrixDataType(name) == Hdf5File::MatrixDataType::kFloat ? 0 : 1;
  *is_complex = f.readMatrixDomainType(name) == Hdf5File::MatrixDomainType::kComplex ? 1 : 0;
  KWH_CATCH
}
/* the same with the 4th (time) extent of per-cuboid series: dims = (x, y, z, t), t = 0 for 3-D datasets */
KWH_API int kwh_h5_dataset_info_4d(const char* path, const char* name, uint64_t dims[4], int32_t* dtype, int32_t* is_complex)
{
  KWH_TRY
  Hdf5File f;
  f.open(path, true);
  const DimensionSizes d = f.getDatasetDimensionSizes(name);
  dims[0] = d.nx; dims[1] = d.ny; dims[2] = d.nz; dims[3] = d.nt;
  *dtype      = f.readMatrixDataType(name) == Hdf5File::MatrixDataType::kFloat ? 0 : 1;
  *is_complex = f.readMatrixDomainType(name) == Hdf5File::MatrixDomainType::kComplex ? 1 : 0;
  KWH_CATCH
}
KWH_API int kwh_h5_read(const char* path, const char* name, void* dst, uint64_t n, int32_t dtype)
{
  KWH_TRY
  Hdf5File f;
  f.open(path, true);
  if (dtype == 0) f.readCompleteDataset(name, n, static_cast<float*>(dst));
  else f.readCompleteDataset(name, n, static_cast<size_t*>(dst));
  KWH_CATCH
}
KWH_API int kwh_h5_read_numeric_attribute(const char* path, const char* dataset, const char* attr, double* out)
{
  KWH_TRY
  Hdf5File f;
  f.open(path, true);
  *out = f.readNumericAttribute(dataset, attr);
  KWH_CATCH
}
KWH_API int kwh_h5_read_attribute(const char* path, const char* dataset, const char* attr, char* out, uint64_t cap)
{
  KWH_TRY
  Hdf5File f;
  f.open(path, true);
  const std::string v = f.readStringAttribute(dataset, attr);
  std::strncpy(out, v.c_str(), cap);
  if (cap) out[cap - 1] = 0;
  KWH_CATCH
}

KWH_API int kwh_write_output_file_ex(kwh_solver* s, const char* path, uint32_t compression_level, int32_t copy_sensor_mask)
{
  KWH_TRY
  KWH_BIND(s)
  if (!s || !path) throw std::invalid_argument("kwh_write_output_file_ex: NULL argument");
  if (compression_level > 9) throw std::invalid_argument("compression level must be 0..9");
  kwh_write_output(s, path, compression_level, copy_sensor_mask != 0);
  KWH_CATCH
}

/* Open the output file before the first step: every stored time series (raw and compressed streams) is then appended
 * to it step by step instead of being held in host memory until the end; kwh_write_output_file(_ex) on the same path
 * completes the file.  reopen != 0 continues the output file of a checkpointed run (call before kwh_checkpoint_read). */
KWH_API int kwh_open_output_file(kwh_solver* s, const char* path, uint32_t compression_level, int32_t reopen)
{
  KWH_TRY
  KWH_BIND(s)
  if (!s || !path) throw std::invalid_argument("kwh_open_output_file: NULL argument");
  kwh_open_output(s, path, compression_level, reopen != 0);
  KWH_CATCH
}

KWH_API int kwh_post_process_output_file(kwh_solver* s, const char* path)
{
  KWH_TRY
  KWH_BIND(s)
  if (!s || !path) throw std::invalid_argument("kwh_post_process_output_file: NULL argument");
  kwh_post_process_output(s, path);
  KWH_CATCH
}

KWH_API int kwh_checkpoint_write(kwh_solver* s, const char* path)
{
  KWH_TRY
  KWH_BIND(s)
  if (!s || !path) throw std::invalid_argument("kwh_checkpoint_write: NULL argument");
  kwh_checkpoint_write_impl(s, path);
  KWH_CATCH
}

KWH_API int kwh_checkpoint_read(kwh_solver* s, const char* path)
{
  KWH_TRY
  KWH_BIND(s)
  if (!s || !path) throw std::invalid_argument("kwh_checkpoint_read: NULL argument");
  kwh_checkpoint_read_impl(s, path);
  KWH_CATCH
}
}
