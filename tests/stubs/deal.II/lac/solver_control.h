// declaration-only stand-in (see ../../README.md)
#pragma once
#include <exception>
namespace dealii {
class SolverControl {
public:
  class NoConvergence : public std::exception {
  public:
    NoConvergence(const unsigned int last_step, const double last_residual);
    const unsigned int last_step;
    const double last_residual;
  };
};
}  // namespace dealii
