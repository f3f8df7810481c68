// Probe: hipExtStreamCreateWithCUMask with full and partial masks on this stack (256 CUs = 8 words of 32 bits).
// Prints the return code of every step; each variant runs in a child process so that a crash in the runtime
// is reported (signal) instead of ending the probe.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

__global__ void touch(int* out) { out[blockIdx.x] = 1; }

static int variant(int words, int ones_words, unsigned partial) {
  std::vector<uint32_t> mask(words, 0u);
  for (int k = 0; k < ones_words && k < words; ++k) mask[k] = 0xffffffffu;
  if (partial && ones_words < words) mask[ones_words] = partial;
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data());
  printf("  create(words=%d, full words=%d, partial=0x%x) -> %d (%s)\n", words, ones_words, partial, (int)e, hipGetErrorString(e));
  fflush(stdout);
  if (e != hipSuccess) return 1;
  int* out = nullptr;
  if (hipMalloc(&out, 4096 * sizeof(int)) != hipSuccess) return 2;
  hipLaunchKernelGGL(touch, dim3(1024), dim3(64), 0, st, out);
  e = hipStreamSynchronize(st);
  printf("  launch + sync -> %d (%s)\n", (int)e, hipGetErrorString(e));
  uint32_t got[16] = {0};
  e = hipExtStreamGetCUMask(st, 8, got);
  printf("  get mask -> %d: %08x %08x %08x %08x %08x %08x %08x %08x\n", (int)e, got[0], got[1], got[2], got[3], got[4], got[5], got[6], got[7]);
  e = hipStreamDestroy(st);
  printf("  destroy -> %d\n", (int)e);
  fflush(stdout);
  return 0;
}

int main() {
  struct { int words, ones; unsigned partial; } v[] = {{8, 8, 0}, {8, 7, 0}, {8, 4, 0}, {8, 7, 0x0000ffffu}, {8, 0, 0x1u}, {1, 1, 0}, {4, 4, 0}};
  hipDeviceProp_t prop;
  for (auto& c : v) {
    printf("variant words=%d full=%d partial=0x%x\n", c.words, c.ones, c.partial);
    fflush(stdout);
    pid_t pid = fork();
    if (pid == 0) {
      if (hipGetDeviceProperties(&prop, 0) != hipSuccess) _exit(9);
      printf("  device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
      _exit(variant(c.words, c.ones, c.partial));
    }
    int status = 0;
    waitpid(pid, &status, 0);
    if (WIFSIGNALED(status)) printf("  child ended by signal %d (%s)\n", WTERMSIG(status), strsignal(WTERMSIG(status)));
    else printf("  child exit code %d\n", WEXITSTATUS(status));
    fflush(stdout);
  }
  return 0;
}
