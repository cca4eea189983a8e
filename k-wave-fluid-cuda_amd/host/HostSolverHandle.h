// HostSolverHandle.h — the object behind the opaque kwh_solver* of include/kwave_host.h (shared by host_capi.cpp and
// the optional HDF5 entry points in h5/h5_capi.cpp).
#ifndef KW_HOST_SOLVER_HANDLE_H
#define KW_HOST_SOLVER_HANDLE_H
#include <memory>
#include <string>

#include "KSpaceFirstOrderSolver.h"
#include "kwave_host.h"

struct kwh_solver
{
  // this solver's parameter set, device context and compression basis; every entry point binds it to the calling
  // thread (KWH_BIND) so that Parameters::getInstance() inside the solver classes means THIS solver's set
  std::unique_ptr<Parameters>             params = Parameters::createDetached();
  MemoryInput                             input;       // datasets handed over in memory (kwh_create)
  std::unique_ptr<InputProvider>          file_input;  // or an input file (kwh_create_from_file)
  std::unique_ptr<KSpaceFirstOrderSolver> solver;
  // HDF5 component: the writer behind kwh_open_output_file (h5/SeriesWriter.h), kept type-erased because this header is
  // shared with the HDF5-free library; must go before the solver (its sinks live in the solver's streams)
  std::shared_ptr<void>                   series_writer;
  ~kwh_solver()
  {
    Parameters::Scope bound(params.get()); // the streams and matrices are released on this solver's device context
    series_writer.reset();
    solver.reset();
  }
};

#define KWH_BIND(s) Parameters::Scope kwhBoundScope((s) != nullptr ? (s)->params.get() : nullptr);

void kwh_set_error(const std::string& e);
Parameters::Options kwh_convert_options(const kwh_options* o);
void kwh_build_solver(kwh_solver& s, const InputProvider& input, const Parameters::Options& opt);
#endif
