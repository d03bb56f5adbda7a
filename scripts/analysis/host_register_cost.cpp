// How long does it take to pin (hipHostRegister) a freshly faulted 1.44 GB result buffer, and how fast is a DMA into it?
// hipcc scripts/analysis/host_register_cost.cpp -o /tmp/hrc && /tmp/hrc      (run on the GPU box; analysis only)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const size_t bytes = (size_t)1440 << 20, huge = (size_t)2 << 20;
	void *d = nullptr;
	hipMalloc(&d, bytes);
	hipMemset(d, 1, bytes);
	for (int rep = 0; rep < 3; rep++) {
		char *base = (char *)mmap(nullptr, bytes + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		char *p = (char *)(((uintptr_t)base + huge - 1) & ~(uintptr_t)(huge - 1));
		madvise(p, bytes, MADV_HUGEPAGE);
		double t0 = now();
		std::vector<std::thread> th;
		for (int t = 0; t < 16; t++) th.emplace_back([=]() { for (size_t o = bytes*t/16; o < bytes*(t + 1)/16; o += 4096) p[o] = 0; });
		for (auto &x : th) x.join();
		double t1 = now();
		hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
		double t2 = now();
		hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost);
		double t3 = now();
		hipHostUnregister(p);
		double t4 = now();
		printf("rep %d: prefault %.1f ms, hipHostRegister %.1f ms (%s), D2H into it %.1f ms = %.1f GB/s, unregister %.1f ms\n", rep, t1 - t0, t2 - t1,
		       hipGetErrorString(e), t3 - t2, bytes/(t3 - t2)/1e6, t4 - t3);
		munmap(base, bytes + huge);
	}
	return 0;
}
