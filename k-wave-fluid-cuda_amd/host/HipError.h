// HipError.h — restores the reference's error convention above the C-ABI: a failing device call becomes a C++
// exception (reference: cudaCheckErrors -> std::runtime_error, Logger/Logger.h:194-216; cuFFT ->
// throwCufftException, CufftComplexMatrix.cpp:706-720; allocation -> std::bad_alloc, BaseFloatMatrix.cpp:140-143).
#ifndef KW_HOST_HIP_ERROR_H
#define KW_HOST_HIP_ERROR_H
#include <new>
#include <stdexcept>

#include "kwave_hip.h"

inline void kwCheck(kw_status s)
{
  if (s == KW_OK) return;
  if (s == KW_ERR_ALLOC) throw std::bad_alloc();
  throw std::runtime_error(kw_last_error());
}
#endif
