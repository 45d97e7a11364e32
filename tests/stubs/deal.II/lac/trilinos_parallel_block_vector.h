// declaration-only stand-in (see ../../README.md)
#pragma once
#include "../base/index_set.h"
namespace dealii {
namespace TrilinosWrappers {
namespace MPI {
class BlockVector {
public:
  using size_type = types::global_dof_index;
  size_type size() const;
  IndexSet locally_owned_elements() const;
  double operator[](const size_type i) const;
  double &operator[](const size_type i);
  void compress(VectorOperation::values operation);
  BlockVector &operator=(const BlockVector &);
};
}  // namespace MPI
}  // namespace TrilinosWrappers
}  // namespace dealii
