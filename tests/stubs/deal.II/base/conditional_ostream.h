// declaration-only stand-in (see ../../README.md)
#pragma once
#include <ostream>
namespace dealii {
class ConditionalOStream {
public:
  template <typename T>
  const ConditionalOStream &operator<<(const T &t) const;
  const ConditionalOStream &operator<<(std::ostream &(*p)(std::ostream &)) const;
};
}  // namespace dealii
