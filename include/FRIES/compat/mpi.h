/* One-rank stand-in for <mpi.h>, for programs written against the reference's headers (which call MPI directly) when they are
 * built against the MI355X engine without an MPI installation: every communicator has size 1, collectives copy.  With a real MPI,
 * put its include directory BEFORE include/FRIES/compat and link it: the same headers then run one rank per GPU -- Adder::perform_add
 * routes the adds with MPI_Alltoallv, sum_mpi gathers with MPI_Allgather, and the engine's own collectives reach MPI through
 * fries_hostcomm_create (include/FRIES/backend.hpp). */
#ifndef FRIES_COMPAT_MPI_H
#define FRIES_COMPAT_MPI_H
#include <stddef.h>
#include <string.h>
typedef int MPI_Comm;
typedef int MPI_Datatype;
#define FRIES_COMPAT_MPI 1
#define MPI_COMM_WORLD 0
#define MPI_SUCCESS 0
#define MPI_IN_PLACE ((void *)1)
enum { MPI_CHAR = 1, MPI_UINT8_T = 1, MPI_BYTE = 1, MPI_INT = 4, MPI_UNSIGNED = 5, MPI_UINT32_T = 5, MPI_DOUBLE = 8, MPI_LONG_LONG = 9, MPI_UINT64_T = 10, MPI_UNSIGNED_LONG = 10 };
static inline size_t fries_mpi_size(MPI_Datatype t) { return t == 1 ? 1 : (t == 4 || t == 5) ? 4 : 8; }
static inline int MPI_Init(int *, char ***) { return 0; }
static inline int MPI_Finalize(void) { return 0; }
static inline int MPI_Comm_size(MPI_Comm, int *n) { *n = 1; return 0; }
static inline int MPI_Comm_rank(MPI_Comm, int *r) { *r = 0; return 0; }
static inline int MPI_Barrier(MPI_Comm) { return 0; }
static inline int MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm) { return 0; }
static inline int MPI_Gather(const void *s, int n, MPI_Datatype t, void *r, int, MPI_Datatype, int, MPI_Comm) { if (s != MPI_IN_PLACE && s != r) memcpy(r, s, n * fries_mpi_size(t)); return 0; }
static inline int MPI_Scatter(const void *s, int n, MPI_Datatype t, void *r, int, MPI_Datatype, int, MPI_Comm) { if (r != MPI_IN_PLACE && s != r) memcpy(r, s, n * fries_mpi_size(t)); return 0; }
static inline int MPI_Allgather(const void *s, int n, MPI_Datatype t, void *r, int, MPI_Datatype, MPI_Comm) { if (s != MPI_IN_PLACE && s != r) memcpy(r, s, n * fries_mpi_size(t)); return 0; }
static inline int MPI_Alltoall(const void *s, int n, MPI_Datatype t, void *r, int, MPI_Datatype, MPI_Comm) { if (s != r) memcpy(r, s, n * fries_mpi_size(t)); return 0; }
static inline int MPI_Alltoallv(const void *s, const int *sc, const int *sd, MPI_Datatype t, void *r, const int *, const int *rd, MPI_Datatype, MPI_Comm) {
    memcpy((char *)r + (size_t)rd[0] * fries_mpi_size(t), (const char *)s + (size_t)sd[0] * fries_mpi_size(t), (size_t)sc[0] * fries_mpi_size(t));
    return 0;
}
#endif
