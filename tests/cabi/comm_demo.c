/* A C host's multi-GPU path: one process per GPU, the error counters of a Monte-Carlo iteration summed over
 * the ranks with ONE all-reduce through the C-ABI (samd_comm_*, RCCL underneath) - no Python, no torch in this process.
 *
 *   comm_demo <rank> <world_size> <id_file>
 *
 * Rank 0 makes the communicator id and writes it to <id_file>; the other ranks wait for the file (the "out of band"
 * channel of include/sionna_amd.h).  Every rank binds GPU (rank mod device count), contributes counters
 * {1, 10, 100, 1000} * (rank + 1) and checks the sum.  tests/test_gpu_cabi_c.py runs it with world_size 1 on the one
 * leased GPU; on a multi-GPU node start world_size copies. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include "sionna_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_SAMD(x) do { int r_ = (x); if (r_ != SAMD_OK) { printf("samd error %d at %s:%d: %s\n", r_, __FILE__, __LINE__, samd_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
  if (argc < 4) { printf("usage: comm_demo <rank> <world_size> <id_file>\n"); return 1; }
  const int rank = atoi(argv[1]), world = atoi(argv[2]);
  const char* id_file = argv[3];
  const int ndev = samd_device_count();
  if (ndev < 1) { printf("no device\n"); return 4; }
  CHECK_HIP(hipSetDevice(rank % ndev));

  unsigned char id[SAMD_COMM_ID_BYTES];
  if (rank == 0) {
    CHECK_SAMD(samd_comm_unique_id(id));
    char tmp[4096];
    snprintf(tmp, sizeof(tmp), "%s.tmp", id_file);
    FILE* f = fopen(tmp, "wb");
    if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) { printf("cannot write %s\n", tmp); return 1; }
    fclose(f);
    if (rename(tmp, id_file) != 0) { printf("cannot rename to %s\n", id_file); return 1; }
  } else {
    FILE* f = NULL;
    for (int tries = 0; tries < 600 && !(f = fopen(id_file, "rb")); ++tries) usleep(100000);
    if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) { printf("cannot read %s\n", id_file); return 1; }
    fclose(f);
  }

  samd_comm_t* comm = NULL;
  CHECK_SAMD(samd_comm_create(id, rank, world, &comm));
  if (samd_comm_rank(comm) != rank || samd_comm_world_size(comm) != world) { printf("bad communicator\n"); return 5; }

  int64_t h[4] = {1, 10, 100, 1000};
  for (int i = 0; i < 4; ++i) h[i] *= (rank + 1);
  int64_t* d = NULL;
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  CHECK_HIP(hipMalloc((void**)&d, sizeof(h)));
  CHECK_HIP(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
  for (int rep = 0; rep < 3; ++rep) {                     /* three iterations: the sum of sums grows by world each time */
    CHECK_SAMD(samd_comm_allreduce_sum_i64(comm, d, 4, st));
    CHECK_HIP(hipStreamSynchronize(st));
  }
  CHECK_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  /* after the first reduce every rank holds S = sum_r (r+1) * base; each further reduce multiplies by world */
  const int64_t tri = (int64_t)world * (world + 1) / 2;
  int64_t scale = tri;
  for (int rep = 1; rep < 3; ++rep) scale *= world;
  const int64_t base[4] = {1, 10, 100, 1000};
  for (int i = 0; i < 4; ++i)
    if (h[i] != base[i] * scale) { printf("rank %d: counter %d = %lld, expected %lld\n", rank, i, (long long)h[i], (long long)(base[i] * scale)); return 6; }
  samd_comm_destroy(comm);
  CHECK_HIP(hipFree(d));
  printf("rank %d of %d: counters %lld %lld %lld %lld\nCOMM_DEMO_OK\n", rank, world, (long long)h[0], (long long)h[1], (long long)h[2], (long long)h[3]);
  return 0;
}
