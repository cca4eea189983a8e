// MatrixNames.h — HDF5 dataset names of the k-Wave file format 1.1 used on the hot path
// (mirror of Utils/MatrixNames.h:41-346 of the reference; names are the wire format).
#ifndef KW_HOST_MATRIX_NAMES_H
#define KW_HOST_MATRIX_NAMES_H
#include <string>

using MatrixName = const std::string;

MatrixName kNtName = "Nt", kNxName = "Nx", kNyName = "Ny", kNzName = "Nz";
MatrixName kDtName = "dt", kDxName = "dx", kDyName = "dy", kDzName = "dz", kCRefName = "c_ref";
MatrixName kC0Name = "c0", kRho0Name = "rho0", kRho0SgxName = "rho0_sgx", kRho0SgyName = "rho0_sgy",
           kRho0SgzName = "rho0_sgz";
MatrixName kBonAName = "BonA", kAlphaCoeffName = "alpha_coeff", kAlphaPowerName = "alpha_power";
MatrixName kPmlXSizeName = "pml_x_size", kPmlYSizeName = "pml_y_size", kPmlZSizeName = "pml_z_size";
MatrixName kPmlXAlphaName = "pml_x_alpha", kPmlYAlphaName = "pml_y_alpha", kPmlZAlphaName = "pml_z_alpha";
MatrixName kPmlXName = "pml_x", kPmlYName = "pml_y", kPmlZName = "pml_z";
MatrixName kPmlXSgxName = "pml_x_sgx", kPmlYSgyName = "pml_y_sgy", kPmlZSgzName = "pml_z_sgz";
MatrixName kDdxKShiftPosRName = "ddx_k_shift_pos_r", kDdyKShiftPosName = "ddy_k_shift_pos",
           kDdzKShiftPosName = "ddz_k_shift_pos";
MatrixName kDdxKShiftNegRName = "ddx_k_shift_neg_r", kDdyKShiftNegName = "ddy_k_shift_neg",
           kDdzKShiftNegName = "ddz_k_shift_neg";
MatrixName kXShiftNegRName = "x_shift_neg_r", kYShiftNegRName = "y_shift_neg_r", kZShiftNegRName = "z_shift_neg_r";
MatrixName kPressureSourceFlagName = "p_source_flag", kInitialPressureSourceFlagName = "p0_source_flag",
           kTransducerSourceFlagName = "transducer_source_flag";
MatrixName kVelocityXSourceFlagName = "ux_source_flag", kVelocityYSourceFlagName = "uy_source_flag",
           kVelocityZSourceFlagName = "uz_source_flag";
MatrixName kNonUniformGridFlagName = "nonuniform_grid_flag", kAbsorbingFlagName = "absorbing_flag",
           kNonLinearFlagName = "nonlinear_flag";
MatrixName kPressureSourceModeName = "p_source_mode", kPressureSourceManyName = "p_source_many";
MatrixName kVelocitySourceModeName = "u_source_mode", kVelocitySourceManyName = "u_source_many";
MatrixName kPressureSourceInputName = "p_source_input", kPressureSourceIndexName = "p_source_index";
MatrixName kVelocitySourceIndexName = "u_source_index", kVelocityXSourceInputName = "ux_source_input",
           kVelocityYSourceInputName = "uy_source_input", kVelocityZSourceInputName = "uz_source_input";
MatrixName kInitialPressureSourceInputName = "p0_source_input";
MatrixName kTransducerSourceInputName = "transducer_source_input", kDelayMaskName = "delay_mask";
MatrixName kSensorMaskTypeName = "sensor_mask_type", kSensorMaskIndexName = "sensor_mask_index",
           kSensorMaskCornersName = "sensor_mask_corners";
// output / state names
MatrixName kPName = "p", kPRmsName = "p_rms", kPMaxName = "p_max", kPMinName = "p_min", kPMaxAllName = "p_max_all",
           kPMinAllName = "p_min_all", kPressureFinalName = "p_final";
MatrixName kUxName = "ux", kUyName = "uy", kUzName = "uz";
MatrixName kUxNonStaggeredName = "ux_non_staggered", kUyNonStaggeredName = "uy_non_staggered",
           kUzNonStaggeredName = "uz_non_staggered";
MatrixName kUxFinalName = "ux_final", kUyFinalName = "uy_final", kUzFinalName = "uz_final";
MatrixName kDxudxnName = "dxudxn", kDyudynName = "dyudyn", kDzudznName = "dzudzn", kDxudxnSgxName = "dxudxn_sgx",
           kDyudynSgyName = "dyudyn_sgy", kDzudznSgzName = "dzudzn_sgz"; // non-uniform grid (MatrixNames.h:100-111)
MatrixName kRhoXName = "rhox", kRhoYName = "rhoy", kRhoZName = "rhoz";
MatrixName kUxSgxName = "ux_sgx", kUySgyName = "uy_sgy", kUzSgzName = "uz_sgz";
MatrixName kTimeIndexName = "t_index";
#endif
