// declaration-only stand-in (see README.md): the MPI names include/nsx_dealii_adaptor.hpp uses
#pragma once
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
typedef int MPI_Request;
struct MPI_Status;
extern MPI_Comm MPI_COMM_SELF, MPI_COMM_WORLD;
extern MPI_Datatype MPI_DOUBLE;
extern MPI_Op MPI_SUM;
#define MPI_IN_PLACE ((void *)1)
#define MPI_STATUSES_IGNORE ((MPI_Status *)0)
int MPI_Allreduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype type, MPI_Op op, MPI_Comm comm);
int MPI_Irecv(void *buf, int count, MPI_Datatype type, int source, int tag, MPI_Comm comm, MPI_Request *request);
int MPI_Isend(const void *buf, int count, MPI_Datatype type, int dest, int tag, MPI_Comm comm, MPI_Request *request);
int MPI_Waitall(int count, MPI_Request *requests, MPI_Status *statuses);
