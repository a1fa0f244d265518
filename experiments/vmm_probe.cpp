// Probe (not product code): which osHandle convention does hipMemImportFromShareableHandle use on this ROCm?
#include <hip/hip_runtime_api.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
int try_form(int form)
{
  pid_t pid = fork();
  if (pid == 0) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) _exit(10);
    hipMemGenericAllocationHandle_t h, h2;
    if (hipMemCreate(&h, gran, &prop, 0) != hipSuccess) _exit(11);
    int fd = -1;
    if (hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0) != hipSuccess) _exit(12);
    hipError_t e = form == 0 ? hipMemImportFromShareableHandle(&h2, (void*)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor)
                             : hipMemImportFromShareableHandle(&h2, (void*)&fd, hipMemHandleTypePosixFileDescriptor);
    if (e != hipSuccess) _exit(20);
    void* va = nullptr;
    if (hipMemAddressReserve(&va, gran, gran, nullptr, 0) != hipSuccess) _exit(21);
    if (hipMemMap(va, gran, 0, h2, 0) != hipSuccess) _exit(22);
    hipMemAccessDesc a{};
    a.location.type = hipMemLocationTypeDevice; a.location.id = 0; a.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, gran, &a, 1) != hipSuccess) _exit(23);
    if (hipMemset(va, 7, gran) != hipSuccess) _exit(24);
    printf("form %d OK (granularity %zu)\n", form, gran);
    _exit(0);
  }
  int st = 0;
  waitpid(pid, &st, 0);
  if (WIFSIGNALED(st)) { printf("form %d: killed by signal %d\n", form, WTERMSIG(st)); return -1; }
  printf("form %d: exit %d\n", form, WEXITSTATUS(st));
  return WEXITSTATUS(st);
}
int main() { try_form(0); try_form(1); return 0; }
