/* kwave_host_h5.h — file-format side of the host layer (optional component libkwave_host_h5.so, needs libhdf5):
 * k-Wave HDF5 input / output / checkpoint files around the solver of kwave_host.h.
 *
 * Replaces, for the path this build covers: main() (main.cpp:840-966: load input, run, write output),
 * Hdf5File / Hdf5FileHeader (Hdf5/Hdf5File.cpp:97-1086, Hdf5FileHeader.cpp:62-200), the output-file scalars
 * (Parameters.cpp:559-647), and checkpoint / restart (KSpaceSolver/KSpaceFirstOrderSolver.cpp:1176-1224 save,
 * :186-228 recover, :1124-1169 file checks; Containers/MatrixContainer.cpp:504-537).
 * Same conventions as kwave_host.h: int status (0 = ok), message through kwh_last_error(). */
#ifndef KWAVE_HOST_H5_H
#define KWAVE_HOST_H5_H
#include "kwave_host.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Parameters::init + loadInputData from a k-Wave input file (file format 1.1; main.cpp:857-917) */
KWH_API int kwh_create_from_file(const char* input_path, const kwh_options* options, kwh_solver** out);
/* output file: header, scalars, one dataset per stream, p_final / u_final (KSpaceFirstOrderSolver.cpp:950-1053) */
KWH_API int kwh_write_output_file(kwh_solver* s, const char* path);
/* the same with the reference's -c <deflate level 0..9> (datasets are chunked as in RealMatrix.cpp:88-121 either way) and
 * --copy_sensor_mask (KSpaceFirstOrderSolver.cpp:1036-1052: sensor_mask_index / sensor_mask_corners, 1-based) */
KWH_API int kwh_write_output_file_ex(kwh_solver* s, const char* path, uint32_t compression_level, int32_t copy_sensor_mask);
/* Per-step output (OutputStreamContainer.cpp:380-403, IndexOutputStream.cpp:87-160, :348-372): open the output file before
 * the first step; every stored time series (raw and compression streams) then owns its dataset(s) in it and appends one
 * hyperslab per sampled step / finished frame through a writer thread, instead of being held in host memory until the
 * end.  kwh_write_output_file(_ex) on the same path completes the file.  reopen != 0 continues the output file of a
 * checkpointed run (call before kwh_checkpoint_read). */
KWH_API int kwh_open_output_file(kwh_solver* s, const char* path, uint32_t compression_level, int32_t reopen);
/* --post (KSpaceFirstOrderSolver.cpp:373-415, :977-1024): for a solver created with kwh_options.only_post_processing —
 * I_avg / Q_term from the p and u_non_staggered series, I_avg_c / Q_term_c from the coefficient frames that an earlier
 * run stored in the output file `path`; the results are added to that file (replacing earlier ones of the same name). */
KWH_API int kwh_post_process_output_file(kwh_solver* s, const char* path);
/* write an input file from in-memory datasets (what the MATLAB side / a generator produces); is_complex[i] != 0 marks
 * interleaved complex float data (domain_type = "complex", fastest dimension doubled: Hdf5File.cpp:898-915) */
KWH_API int kwh_write_input_file(const char* path, const kwh_dataset* datasets, size_t n, const int32_t* is_complex);
/* reading back: dims (x,y,z), dtype (0 float / 1 uint64), domain; whole dataset; string attribute ("/" = root) */
KWH_API int kwh_h5_dataset_info(const char* path, const char* name, uint64_t dims[3], int32_t* dtype, int32_t* is_complex);
/* dims = (x, y, z, t); t = 0 for a 3-D dataset, > 0 for the per-cuboid series "/<stream>/<cuboid>" of a corners mask */
KWH_API int kwh_h5_dataset_info_4d(const char* path, const char* name, uint64_t dims[4], int32_t* dtype, int32_t* is_complex);
KWH_API int kwh_h5_read(const char* path, const char* name, void* dst, uint64_t n, int32_t dtype);
KWH_API int kwh_h5_dataset_exists(const char* path, const char* name, int32_t* exists);
/* planes [z0, z0 + n_planes) of a 3-D float dataset: the part of a grid-sized input array one slab rank needs */
KWH_API int kwh_h5_read_planes(const char* path, const char* name, uint64_t z0, uint64_t n_planes, float* dst);
/* one cuboid of a corner-mask stream into an existing output file: "<group>/<index>", dims (nx, ny, nz, nt); nt = 0: aggregate */
KWH_API int kwh_h5_append_cuboid(const char* path, const char* group, uint64_t index, const uint64_t dims[4], const float* data);
/* file of any type ("input", "output") from in-memory datasets; kwh_write_input_file is this with type "input" */
KWH_API int kwh_write_file(const char* path, const char* file_type, const char* description, const kwh_dataset* sets,
                           size_t n, const int32_t* is_complex);
KWH_API int kwh_h5_read_attribute(const char* path, const char* dataset, const char* attr, char* out, uint64_t cap);
/* integer or float attribute (the compression parameters c_harmonics, c_period, ... of a coefficient dataset) */
KWH_API int kwh_h5_read_numeric_attribute(const char* path, const char* dataset, const char* attr, double* out);
/* checkpoint file (file_type = "checkpoint"): p, rhox, rhoy, rhoz, ux_sgx, uy_sgy, uz_sgz, t_index, Nx, Ny, Nz as in the
 * reference, plus the state of every output stream (stream_<name>, stream_<name>_steps).  kwh_checkpoint_read refuses a
 * file whose type or dimensions do not match (KSpaceFirstOrderSolver.cpp:1132-1168) and leaves the solver at the stored
 * time index; kwh_run then continues bit-identically to an uninterrupted run. */
KWH_API int kwh_checkpoint_write(kwh_solver* s, const char* path);
KWH_API int kwh_checkpoint_read(kwh_solver* s, const char* path);

#ifdef __cplusplus
}
#endif
#endif
