// declaration-only stand-in (see ../../README.md)
#pragma once
#include "../base/types.h"
namespace dealii {
template <int dim>
class FiniteElement {
public:
  const unsigned int dofs_per_cell = 0;
  const FiniteElement<dim> &base_element(const unsigned int index) const;
  double shape_value(const unsigned int i, const Point<dim> &p) const;
  Tensor<1, dim> shape_grad(const unsigned int i, const Point<dim> &p) const;
};
}  // namespace dealii
