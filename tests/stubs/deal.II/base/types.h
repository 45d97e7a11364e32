// declaration-only stand-in (see ../../README.md)
#pragma once
namespace dealii {
namespace types {
using global_dof_index = unsigned int;
}
template <int dim>
class Point {
public:
  double operator[](unsigned int d) const;
};
template <int rank, int dim>
class Tensor {
public:
  double operator[](unsigned int d) const;
};
struct VectorOperation {
  enum values { unknown, insert, add };
};
}  // namespace dealii
