// declaration-only stand-in (see ../../README.md)
#pragma once
#include <vector>

#include "types.h"
namespace dealii {
template <int dim>
class Quadrature {
public:
  unsigned int size() const;
  const Point<dim> &point(const unsigned int i) const;
  const std::vector<double> &get_weights() const;
};
}  // namespace dealii
