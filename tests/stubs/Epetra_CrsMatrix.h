// declaration-only stand-in (see README.md)
#pragma once
#include "Epetra_Map.h"
class Epetra_CrsMatrix {
public:
  int ExtractCrsDataPointers(int *&row_offsets, int *&local_columns, double *&values) const;
  int NumMyRows() const;
  const Epetra_Map &ColMap() const;
};
