// DimensionSizes.h — POD of grid dimensions (mirror of Utils/DimensionSizes.h:69-171 of the reference).
#ifndef KW_HOST_DIMENSION_SIZES_H
#define KW_HOST_DIMENSION_SIZES_H
#include <cstddef>

struct DimensionSizes
{
  size_t nx = 0, ny = 0, nz = 0, nt = 0;
  DimensionSizes() = default;
  DimensionSizes(size_t x, size_t y, size_t z, size_t t = 0) : nx(x), ny(y), nz(z), nt(t) {}
  size_t nElements() const { return (nt > 0) ? nx * ny * nz * nt : nx * ny * nz; }
  bool   is2D() const { return nz == 1 && nt <= 1; }
  bool   is3D() const { return nz > 1 && nt <= 1; }
  bool   operator==(const DimensionSizes& o) const { return nx == o.nx && ny == o.ny && nz == o.nz && nt == o.nt; }
  bool   operator!=(const DimensionSizes& o) const { return !(*this == o); }
};
#endif
