// declaration-only stand-in (see ../../README.md)
#pragma once
#include <vector>

#include "../base/types.h"
namespace dealii {
template <int dim>
class DoFCellAccessor {
public:
  bool is_locally_owned() const;
  void get_dof_indices(std::vector<types::global_dof_index> &dof_indices) const;
  Point<dim> &vertex(const unsigned int i) const;
};
template <int dim>
class DoFHandler {
public:
  class active_cell_iterator {
  public:
    const DoFCellAccessor<dim> *operator->() const;
    const active_cell_iterator &operator*() const;   // a range-for over active_cell_iterators() yields iterators
    active_cell_iterator &operator++();
    bool operator!=(const active_cell_iterator &) const;
  };
  struct IteratorRange {
    struct It {
      const active_cell_iterator &operator*() const;
      It &operator++();
      bool operator!=(const It &) const;
    };
    It begin() const;
    It end() const;
  };
  IteratorRange active_cell_iterators() const;
};
}  // namespace dealii
