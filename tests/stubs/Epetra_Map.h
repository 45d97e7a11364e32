// declaration-only stand-in (see README.md)
#pragma once
class Epetra_BlockMap {
public:
  int GID(int local_id) const;
};
class Epetra_Map : public Epetra_BlockMap {};
