// declaration-only stand-in (see ../../README.md)
#pragma once
#include <map>
#include <vector>

#include <mpi.h>
namespace dealii {
namespace Utilities {
namespace MPI {
unsigned int this_mpi_process(const MPI_Comm &comm);
unsigned int n_mpi_processes(const MPI_Comm &comm);
template <typename T>
std::vector<T> all_gather(const MPI_Comm &comm, const T &object_to_send);
template <typename T>
std::map<unsigned int, T> some_to_some(const MPI_Comm &comm, const std::map<unsigned int, T> &objects_to_send);
}  // namespace MPI
}  // namespace Utilities
}  // namespace dealii
