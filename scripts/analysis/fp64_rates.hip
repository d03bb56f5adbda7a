/* Issue cost and accuracy of the fp64 instructions the Fresnel sweep is made of (gfx950).
 *   hipcc -O3 --offload-arch=gfx950 scripts/analysis/fp64_rates.hip -o /tmp/fp64_rates && /tmp/fp64_rates
 * One wave per SIMD (grid 256 x 256 threads) and four (256 x 1024): cycles per wave-instruction from s_memtime. */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define N_IT 4096

template <int OP, int CHAINS>
__global__ void rate(double *out, double seed)
{
	double x[CHAINS];
	for (int c = 0; c < CHAINS; c++) x[c] = seed + 1e-3*threadIdx.x + c;
	long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < N_IT; it++) {
#pragma unroll
		for (int c = 0; c < CHAINS; c++) {
			if (OP == 0) x[c] = __builtin_fma(x[c], 1.0000001, 1e-9);
			if (OP == 1) x[c] = __builtin_amdgcn_rsq(x[c]) + 2.0;
			if (OP == 2) x[c] = __builtin_amdgcn_rcp(x[c]) + 2.0;
			if (OP == 3) x[c] = __builtin_amdgcn_sqrt(x[c]) + 2.0;
			if (OP == 4) x[c] = x[c] * 1.0000001;
			if (OP == 5) x[c] = x[c] + 1e-9;
			if (OP == 6) x[c] = sqrt(x[c]) + 2.0;              /* correctly rounded sequence */
			if (OP == 7) x[c] = 3.0 / x[c] + 2.0;              /* correctly rounded division */
			if (OP == 8) x[c] = __builtin_amdgcn_ldexp(x[c], 1) * 0.5;
			if (OP == 9) x[c] = (x[c] > 2.5) ? x[c] - 1.0 : x[c] + 0.5;
		}
	}
	long long t1 = __builtin_readcyclecounter();
	double s = 0;
	for (int c = 0; c < CHAINS; c++) s += x[c];
	out[blockIdx.x*blockDim.x + threadIdx.x] = s + (double)(t1 - t0)*0.0;
	if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}

__global__ void acc(const double *in, double *o_rsq, double *o_rcp, double *o_sqrt, int n)
{
	int i = blockIdx.x*blockDim.x + threadIdx.x;
	if (i >= n) return;
	o_rsq[i] = __builtin_amdgcn_rsq(in[i]);
	o_rcp[i] = __builtin_amdgcn_rcp(in[i]);
	o_sqrt[i] = __builtin_amdgcn_sqrt(in[i]);
}

template <int OP, int CHAINS>
static void run(const char *name, int extra_per_it)
{
	double *d;
	hipMalloc(&d, 256*1024*sizeof(double));
	for (int block : {256, 1024}) {
		hipEvent_t e0, e1;
		hipEventCreate(&e0); hipEventCreate(&e1);
		hipLaunchKernelGGL((rate<OP, CHAINS>), dim3(256), dim3(block), 0, 0, d, 1.5);
		hipEventRecord(e0);
		hipLaunchKernelGGL((rate<OP, CHAINS>), dim3(256), dim3(block), 0, 0, d, 1.5);
		hipEventRecord(e1);
		hipDeviceSynchronize();
		float ms; hipEventElapsedTime(&ms, e0, e1);
		double cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
		const double waves_per_simd = block/256.0;
		const double insts = (double)N_IT*CHAINS*(1 + extra_per_it);
		printf("%-28s chains %d waves/SIMD %.0f: %7.3f ms, wall-clock ns per wave-op-group on a SIMD %.2f (= %.1f clk at 2.4 GHz), counter %.0f\n",
		       name, CHAINS, waves_per_simd, ms, ms*1e6/(insts*waves_per_simd)*1.0, ms*1e6/(insts*waves_per_simd)*2.4, cyc);
	}
	hipFree(d);
}

int main()
{
	run<0, 1>("fma dependent", 0); run<0, 8>("fma x8", 0);
	run<4, 8>("mul x8", 0); run<5, 8>("add x8", 0);
	run<1, 1>("rsq+add dependent", 1); run<1, 8>("rsq+add x8", 1);
	run<2, 8>("rcp+add x8", 1); run<3, 8>("sqrt(hw)+add x8", 1);
	run<6, 8>("sqrt(IEEE)+add x8", 0); run<7, 8>("div(IEEE)+add x8", 0);
	run<8, 8>("ldexp+mul x8", 1); run<9, 8>("cmp+2cndmask+.. x8", 0);
	/* accuracy */
	const int n = 1 << 20;
	std::vector<double> h(n), a(n), b(n), c(n);
	srand(1);
	for (int i = 0; i < n; i++) h[i] = ldexp(1.0 + (double)rand()/RAND_MAX, (rand() % 80) - 40);
	double *din, *d1, *d2, *d3;
	hipMalloc(&din, n*8); hipMalloc(&d1, n*8); hipMalloc(&d2, n*8); hipMalloc(&d3, n*8);
	hipMemcpy(din, h.data(), n*8, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(acc, dim3(n/256), dim3(256), 0, 0, din, d1, d2, d3, n);
	hipMemcpy(a.data(), d1, n*8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, n*8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d3, n*8, hipMemcpyDeviceToHost);
	double e1 = 0, e2 = 0, e3 = 0;
	for (int i = 0; i < n; i++) {
		long double x = h[i];
		e1 = fmax(e1, (double)fabsl(a[i]*sqrtl(x) - 1.0L));
		e2 = fmax(e2, (double)fabsl(b[i]*x - 1.0L));
		e3 = fmax(e3, (double)fabsl(c[i]/sqrtl(x) - 1.0L));
	}
	printf("max relative error: v_rsq_f64 %.3e (2^%.1f)  v_rcp_f64 %.3e (2^%.1f)  v_sqrt_f64 %.3e (2^%.1f)\n",
	       e1, log2(e1), e2, log2(e2), e3, log2(e3));
	return 0;
}
