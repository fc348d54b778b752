// how long do large hipMalloc calls take right after another process has freed most of the HBM?  (the cold-hipMalloc cliff)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <thread>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t GB = 1ull << 30;
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    double t0 = now();
    hipFree(0);
    printf("context %.3f s\n", now() - t0);
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    printf("free %.1f of %.1f GB\n", fr / 1e9, tot / 1e9);
    if (mode == 0) { // hog: allocate 200 GB, touch it, exit (the next process sees freshly freed memory)
        std::vector<void *> v;
        for (int i = 0; i < 20; i++) { void *p = nullptr; if (hipMalloc(&p, 10 * GB) != hipSuccess) break; hipMemset(p, 1, 10 * GB); v.push_back(p); }
        hipDeviceSynchronize();
        printf("hog: %zu x 10 GB touched\n", v.size());
        return 0;
    }
    // probe: 12 x 10 GB one after the other, then free, then again; then one 120 GB block
    for (int round = 0; round < 2; round++) {
        std::vector<void *> v;
        double ts = now();
        for (int i = 0; i < 12; i++) { void *p = nullptr; double a = now(); hipError_t e = hipMalloc(&p, 10 * GB); printf("round %d malloc %d: %.3f s %s\n", round, i, now() - a, e == hipSuccess ? "" : "FAILED"); if (e == hipSuccess) v.push_back(p); }
        printf("round %d: %zu blocks in %.3f s\n", round, v.size(), now() - ts);
        double tf = now();
        for (void *p : v) hipFree(p);
        printf("round %d: freed in %.3f s\n", round, now() - tf);
    }
    { void *p = nullptr; double a = now(); hipError_t e = hipMalloc(&p, 120 * GB); printf("one 120 GB block: %.3f s %s\n", now() - a, e == hipSuccess ? "" : "FAILED"); double b = now(); hipMemsetAsync(p, 0, 120 * GB, 0); hipDeviceSynchronize(); printf("memset of it: %.3f s\n", now() - b); hipFree(p); }
    return 0;
}
