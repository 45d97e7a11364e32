// declaration-only stand-in (see ../../README.md)
#pragma once
#include "types.h"
namespace dealii {
class IndexSet {
public:
  using size_type = types::global_dof_index;
  class ElementIterator {
  public:
    size_type operator*() const;
    ElementIterator &operator++();
    bool operator!=(const ElementIterator &) const;
  };
  size_type size() const;
  size_type n_elements() const;
  bool is_contiguous() const;
  size_type nth_index_in_set(size_type local_index) const;
  ElementIterator begin() const;
  ElementIterator end() const;
};
}  // namespace dealii
