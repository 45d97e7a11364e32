// declaration-only stand-in (see ../../README.md)
#pragma once
#include "trilinos_sparse_matrix.h"
namespace dealii {
namespace TrilinosWrappers {
class BlockSparseMatrix {
public:
  SparseMatrix &block(const unsigned int row, const unsigned int column);
  const SparseMatrix &block(const unsigned int row, const unsigned int column) const;
};
}  // namespace TrilinosWrappers
}  // namespace dealii
