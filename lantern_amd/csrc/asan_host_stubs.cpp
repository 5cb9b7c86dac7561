// asan_host_stubs.cpp -- only linked into liblantern_hip_asan.so (`make asan`): the HOST side of the library under AddressSanitizer + UBSan
// in the build container (no GPU there, and no GPU AddressSanitizer on the MI355X pool).  The sources are compiled with --offload-host-only, so
// there is no device code object to register: the module constructors' registration calls land on these no-ops (the link uses -Bsymbolic, the
// HIP runtime's own symbols stay untouched for the rest of the process) and every kernel launch reports "no device".  What runs under the
// sanitizers is everything in front of a launch: the host tree builders (tree_static.cpp), lantern_verify_step's sequencing / validation,
// the argument checks and shape arithmetic of every launcher -- tests/test_sanitizers_cpu.py runs tests/test_cabi_cpu.py against it.
#include <hip/hip_runtime.h>

extern "C" {
void **__hipRegisterFatBinary(const void *) {
    static void *handle[2] = {nullptr, nullptr};
    return handle;
}
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned int, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, char *, int, size_t, int, int) {}
void __hipRegisterManagedVar(void *, void **, void *, const char *, size_t, unsigned) {}
}
