// main.cpp — command-line front end "kspaceFirstOrder-HIP": input file -> simulation on an MI355X -> output file.
// Mirrors the sequence of the reference's main() (main.cpp:840-966) and the subset of its flags that select what
// the loop computes and stores (CommandLineParameters.cpp:264-292); cosmetics (usage box, progress table) are out of scope.
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../HipError.h"
#include "../HostSolverHandle.h"
#include "Hdf5File.h"

void kwh_write_output(kwh_solver* s, const std::string& path, unsigned compressionLevel, bool copySensorMask);
void kwh_open_output(kwh_solver* s, const std::string& path, unsigned compressionLevel, bool reopen);
void kwh_post_process_output(kwh_solver* s, const std::string& path);
void kwh_checkpoint_write_impl(kwh_solver* s, const std::string& path);
void kwh_checkpoint_read_impl(kwh_solver* s, const std::string& path);

static void usage()
{
  std::printf("kspaceFirstOrder-HIP -i <input.h5> -o <output.h5> [-g dev] [-s start (1-based)] [--benchmark N]\n"
              "  [-p|--p_raw] [--p_rms] [--p_max] [--p_min] [--p_max_all] [--p_min_all] [--p_final]\n"
              "  [-u|--u_raw] [--u_rms] [--u_max] [--u_min] [--u_max_all] [--u_min_all] [--u_final] [--u_non_staggered_raw]\n"
              "  [--p_c] [--u_c] [--u_non_staggered_c] [--I_avg_c] [--I_avg] [--Q_term] [--Q_term_c]\n"
              "  [--period P | --frequency F] [--mos M] [--harmonics H] [--no_overlap] [--granular]\n"
              "  [-c <deflate 0..9>] [--copy_sensor_mask]\n"
              "  [--checkpoint_file <ckpt.h5> --checkpoint_timesteps N | --checkpoint_interval SECONDS]  stop after N steps\n"
              "      (or once SECONDS of wall-clock time have passed), leaving a checkpoint; the same command line resumes\n"
              "      from it (CommandLineParameters.cpp:264-292)\n"
              "  [--version] [-h|--help]; accepted without effect: -r <percent>, -t <threads>, --verbose <level>,\n"
              "      --block_size <n>\n"
              "  [--40-bit_complex]  compression coefficients kept and stored as 5-byte complex numbers\n"
              "  [--post --I_avg|--I_avg_c|--Q_term|--Q_term_c]  no simulation: compute these from the series stored in <output.h5>\n");
}

// numeric command-line values: anything but a plain non-negative number is an error (the reference's getopt loop
// rejects them the same way, CommandLineParameters.cpp:300-420)
static unsigned long long parseCount(const char* flag, const char* text)
{
  char* end = nullptr;
  errno = 0;
  const unsigned long long v = std::strtoull(text, &end, 10);
  if (text[0] == '-' || text[0] == '\0' || end == text || *end != '\0' || errno != 0)
  {
    std::fprintf(stderr, "Error: %s needs a non-negative integer, got \"%s\"\n", flag, text);
    std::exit(EXIT_FAILURE);
  }
  return v;
}
static double parseReal(const char* flag, const char* text)
{
  char* end = nullptr;
  errno = 0;
  const double v = std::strtod(text, &end);
  if (text[0] == '\0' || end == text || *end != '\0' || errno != 0 || !(v >= 0.0))
  {
    std::fprintf(stderr, "Error: %s needs a non-negative number, got \"%s\"\n", flag, text);
    std::exit(EXIT_FAILURE);
  }
  return v;
}

int main(int argc, char** argv)
{
  std::string in, out, ckpt;
  size_t ckptSteps = 0;
  double ckptSeconds = 0.0;
  unsigned compressionLevel = 0; // -c (CommandLineParameters.h:791: default 0)
  bool copySensorMask = false;
  kwh_options o{};
  o.device_idx    = -1;
  o.fused_kernels = 1;
  o.mos = o.harmonics = 1;
  for (int i = 1; i < argc; i++)
  {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); std::exit(EXIT_FAILURE); } return argv[++i]; };
    if (a == "-i") in = next();
    else if (a == "-o") out = next();
    else if (a == "-g") o.device_idx = static_cast<int32_t>(parseCount("-g", next()));
    else if (a == "-s")
    { // 1-based on the CLI (:416-425); 0 is rejected here, a start beyond Nt by Parameters::init once Nt is known
      const unsigned long long v = parseCount("-s", next());
      if (v < 1) { std::fprintf(stderr, "Error: The beginning of data sampling is out of the simulation time span <1, Nt>.\n"); return EXIT_FAILURE; }
      o.sampling_start_time_index = v - 1;
    }
    else if (a == "--benchmark") o.benchmark_time_steps = parseCount("--benchmark", next());
    else if (a == "-p" || a == "--p_raw") o.p_raw = 1;
    else if (a == "--p_rms") o.p_rms = 1;
    else if (a == "--p_max") o.p_max = 1;
    else if (a == "--p_min") o.p_min = 1;
    else if (a == "--p_max_all") o.p_max_all = 1;
    else if (a == "--p_min_all") o.p_min_all = 1;
    else if (a == "--p_final") o.p_final = 1;
    else if (a == "-u" || a == "--u_raw") o.u_raw = 1;
    else if (a == "--u_rms") o.u_rms = 1;
    else if (a == "--u_max") o.u_max = 1;
    else if (a == "--u_min") o.u_min = 1;
    else if (a == "--u_max_all") o.u_max_all = 1;
    else if (a == "--u_min_all") o.u_min_all = 1;
    else if (a == "--u_final") o.u_final = 1;
    else if (a == "--u_non_staggered_raw") o.u_non_staggered_raw = 1;
    else if (a == "--p_c") o.p_c = 1;
    else if (a == "--u_non_staggered_c") o.u_non_staggered_c = 1;
    else if (a == "--I_avg_c") o.i_avg_c = 1;
    else if (a == "--I_avg") o.i_avg = 1;
    else if (a == "--u_c") o.u_c = 1;
    else if (a == "--frequency") o.frequency = static_cast<float>(parseReal("--frequency", next()));
    else if (a == "--Q_term") o.q_term = 1;
    else if (a == "--Q_term_c") o.q_term_c = 1;
    else if (a == "--period") o.period = static_cast<float>(parseReal("--period", next()));
    else if (a == "--mos") o.mos = parseCount("--mos", next());
    else if (a == "--harmonics") o.harmonics = parseCount("--harmonics", next());
    else if (a == "--no_overlap") o.no_overlap = 1;
    else if (a == "--granular") o.fused_kernels = 0;
    else if (a == "-c") compressionLevel = static_cast<unsigned>(parseCount("-c", next()));
    else if (a == "--copy_sensor_mask") copySensorMask = true;
    else if (a == "--checkpoint_file") ckpt = next();
    else if (a == "--checkpoint_timesteps") ckptSteps = parseCount("--checkpoint_timesteps", next());
    else if (a == "--checkpoint_interval") ckptSeconds = parseReal("--checkpoint_interval", next());
    else if (a == "--version") { std::printf("%s\n", KSpaceFirstOrderSolver().getCodeName().c_str()); return EXIT_SUCCESS; }
    else if (a == "-r" || a == "-t" || a == "--verbose" || a == "--block_size") (void)next(); // progress / threads / log level / host block: nothing to set here
    else if (a == "--post") o.only_post_processing = 1;
    else if (a == "--40-bit_complex") o.complex_40bit = 1;
    else if (a == "-h" || a == "--help") { usage(); return EXIT_SUCCESS; }
    else { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); usage(); return EXIT_FAILURE; }
  }
  if (in.empty() || out.empty()) { usage(); return EXIT_FAILURE; }
  if (o.only_post_processing)
  { // CommandLineParameters.cpp:919-936: --post goes with the post-processed quantities only
    const bool other = o.p_raw || o.p_rms || o.p_max || o.p_min || o.p_max_all || o.p_min_all || o.p_final || o.u_raw || o.u_rms ||
                       o.u_max || o.u_min || o.u_max_all || o.u_min_all || o.u_final || o.u_non_staggered_raw || o.p_c || o.u_c ||
                       o.u_non_staggered_c || !ckpt.empty();
    if (other || !(o.i_avg || o.i_avg_c || o.q_term || o.q_term_c))
    {
      std::fprintf(stderr, "Error: --post takes --I_avg, --I_avg_c, --Q_term, --Q_term_c (at least one) and no other output or checkpoint flag\n");
      return EXIT_FAILURE;
    }
  }
  try
  {
    kwh_solver s;
    KWH_BIND(&s) // the solver classes below find this run's parameter set through Parameters::getInstance()
    s.file_input.reset(new Hdf5Input(in));
    kwh_build_solver(s, *s.file_input, kwh_convert_options(&o));
    s.file_input.reset();
    std::printf("%s on %s\n", s.solver->getCodeName().c_str(),
                Parameters::getInstance().getHipParameters().getDeviceName().c_str());
    Parameters& params = Parameters::getInstance();
    if (o.only_post_processing)
    {
      kwh_post_process_output(&s, out);
      std::printf("post-processing of %s done\n", out.c_str());
      s.solver.reset();
      return EXIT_SUCCESS;
    }
    const bool checkpointing = !ckpt.empty() && (ckptSteps > 0 || ckptSeconds > 0.0);
    bool resuming = false;
    if (checkpointing)
      if (FILE* f = std::fopen(ckpt.c_str(), "rb")) { std::fclose(f); resuming = true; }
    // the output file is open for the whole run: sampled series go to it step by step (OutputStreamContainer.cpp:380-403)
    kwh_open_output(&s, out, compressionLevel, resuming);
    if (checkpointing)
    { // recover if a checkpoint exists (KSpaceFirstOrderSolver.cpp:186-228), run one leg, stop with a new checkpoint
      if (resuming)
      {
        kwh_checkpoint_read_impl(&s, ckpt);
        std::printf("recovered from %s at time step %zu\n", ckpt.c_str(), params.getTimeIndex());
      }
      if (ckptSeconds > 0.0)
      { // --checkpoint_interval: step until the leg has used its wall-clock budget (isTimeToCheckpoint, :1100-1117)
        const auto   t0    = std::chrono::steady_clock::now();
        const size_t limit = (ckptSteps > 0) ? params.getTimeIndex() + ckptSteps : params.getNt();
        while (params.getTimeIndex() < std::min(limit, params.getNt()))
        {
          s.solver->runTimeSteps(1);
          kwCheck(kw_sync(params.getHipParameters().getContext()));
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= ckptSeconds) break;
        }
      }
      else s.solver->runTimeSteps(ckptSteps);
      kwCheck(kw_sync(params.getHipParameters().getContext()));
      if (params.getTimeIndex() < params.getNt())
      {
        kwh_checkpoint_write_impl(&s, ckpt);
        std::printf("time steps: %zu of %zu, checkpoint: %s\n", params.getTimeIndex(), params.getNt(), ckpt.c_str());
        s.solver.reset();
        return EXIT_SUCCESS;
      }
      s.solver->finish();
      std::remove(ckpt.c_str()); // main.cpp:951-958: the checkpoint is deleted once the simulation is complete
    }
    else
    {
      s.solver->compute();
    }
    kwCheck(kw_sync(params.getHipParameters().getContext()));
    kwh_write_output(&s, out, compressionLevel, copySensorMask);
    std::printf("time steps: %zu, output: %s\n", params.getTimeIndex(), out.c_str());
    s.solver.reset();
  }
  catch (const std::exception& e)
  { // Logger::errorAndTerminate (Logger.cpp:82-89)
    std::fprintf(stderr, "Error: %s\n", e.what());
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
