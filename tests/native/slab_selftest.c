/* slab_selftest.c — a plain C driver of the multi-GPU code path, no interpreter anywhere: one rank, Z-slab mode, the
 * all-to-all of every 3-D FFT going through the device library's own RCCL exchange (kw_comm_init, csrc/kw_comm.hip)
 * with itself.  Mirrors the reference's main() sequence (main.cpp:840-966): input file -> time loop -> output file.
 *
 *   slab_selftest <input.h5> <output.h5>
 *
 * Built and run by tests/test_gpu_dist.py::test_native_slab_driver_without_interpreter, which compares the output
 * file with a single-GPU (non-slab) run of the same input. */
#include <stdio.h>
#include <stdlib.h>

#include "kwave_hip.h"
#include "kwave_host.h"

int kwh_create_from_file(const char* input_path, const kwh_options* o, kwh_solver** out);
int kwh_write_output_file(kwh_solver* s, const char* path);
int kwh_h5_read(const char* path, const char* name, void* dst, uint64_t n, int32_t dtype);

#define CHECK(call)                                                                                                    \
  do {                                                                                                                 \
    if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, kwh_last_error()); return EXIT_FAILURE; }             \
  } while (0)

int main(int argc, char** argv)
{
  if (argc != 3) { fprintf(stderr, "usage: %s <input.h5> <output.h5>\n", argv[0]); return EXIT_FAILURE; }
  uint64_t nz = 0, nt = 0;
  CHECK(kwh_h5_read(argv[1], "Nz", &nz, 1, 1));
  CHECK(kwh_h5_read(argv[1], "Nt", &nt, 1, 1));

  unsigned char id[KW_COMM_ID_BYTES];
  if (kw_comm_unique_id(id, sizeof(id)) != KW_OK) { fprintf(stderr, "kw_comm_unique_id: %s\n", kw_last_error()); return EXIT_FAILURE; }

  kwh_options o = {0};
  o.device_idx     = -1;
  o.fused_kernels  = 1;
  o.mos = o.harmonics = 1;
  o.p_raw = o.p_max = o.p_final = o.u_final = 1;
  o.slab_ranks     = 1;   /* this process owns every plane ... */
  o.slab_rank      = 0;
  o.nz_global      = nz;
  o.comm_unique_id = id;  /* ... and still transposes through RCCL, like a rank of a multi-GPU run */

  kwh_solver* s = NULL;
  CHECK(kwh_create_from_file(argv[1], &o, &s));
  CHECK(kwh_run(s, nt));
  CHECK(kwh_finish(s));
  uint64_t exchanges = 0;
  if (kw_comm_info((kw_ctx*)kwh_context(s), NULL, NULL, &exchanges) != KW_OK) return EXIT_FAILURE;
  float fused = 0.0f;
  CHECK(kwh_get_scalar(s, "fused_pipeline", &fused));
  CHECK(kwh_write_output_file(s, argv[2]));
  printf("steps %llu exchanges %llu fused_pipeline %.0f\n", (unsigned long long)nt, (unsigned long long)exchanges, fused);
  CHECK(kwh_destroy(s));
  return EXIT_SUCCESS;
}
