// declaration-only stand-in (see ../../README.md)
#pragma once
#include <Epetra_CrsMatrix.h>
namespace dealii {
namespace TrilinosWrappers {
class SparseMatrix {
public:
  const Epetra_CrsMatrix &trilinos_matrix() const;
};
}  // namespace TrilinosWrappers
}  // namespace dealii
