// atomic_bench.hip -- what does a counting sort by global atomics cost on this device?
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/atomic_bench tools/atomic_bench.hip && tools/build/atomic_bench
//
// The shadow tracer's candidate pairs (3 M of them, ~17 K (beam, size) buckets on the bench frame) only have to be
// GROUPED by bucket: any order inside a bucket gives the same shadow flags.  A radix sort of the pairs costs two
// passes and a histogram kernel; the alternative is count (one atomic per pair or per run of equal keys), scan,
// scatter (one atomic with return per pair or run).  This prints the time of both atomic kernels for random keys and
// for keys in runs (the cull pass emits a beam's pairs together), naive and with one atomic per run of a wave.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32;
#define CHECK(x)                                                                                    \
	do {                                                                                        \
		hipError_t e = (x);                                                                 \
		if (e != hipSuccess) {                                                              \
			fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
			exit(1);                                                                    \
		}                                                                                   \
	} while (0)

__global__ void k_count_naive(const u32 *__restrict__ keys, u32 n, u32 *cnt)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		atomicAdd(&cnt[keys[i]], 1u);
}

// one atomic per run of equal keys among the 64 lanes of a wave
__device__ __forceinline__ void d_runs(u32 key, bool ok, u32 lane, bool &head, u32 &len, u32 &pos)
{
	const u32 prev = (u32)__shfl_up((int)key, 1);
	const bool okp = (bool)__shfl_up((int)ok, 1);
	head = ok && (lane == 0u || !okp || prev != key);
	const unsigned long long heads = __ballot(head), act = __ballot(ok);
	const unsigned long long below = heads & ((2ull << lane) - 1ull);         // heads at or below this lane
	const u32 h = 63u - (u32)__builtin_clzll(below | 1ull);                   // this lane's run head
	const unsigned long long above = heads & ~((2ull << h) - 1ull);           // heads after it
	const u32 end = above ? (u32)__builtin_ctzll(above) : 64u - (u32)__builtin_clzll(act | 1ull);
	len = end - h;
	pos = lane - h;
}

__global__ void k_count_runs(const u32 *__restrict__ keys, u32 n, u32 *cnt)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
	const bool ok = i < n;
	const u32 key = ok ? keys[i] : 0u;
	bool head;
	u32 len, pos;
	d_runs(key, ok, lane, head, len, pos);
	if (head)
		atomicAdd(&cnt[key], len);
}

__global__ void k_scatter_naive(const u32 *__restrict__ keys, const u32 *__restrict__ vals, u32 n, const u32 *__restrict__ start, u32 *cur,
				u32 *out)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		const u32 k = keys[i];
		out[start[k] + atomicAdd(&cur[k], 1u)] = vals[i];
	}
}

__global__ void k_scatter_runs(const u32 *__restrict__ keys, const u32 *__restrict__ vals, u32 n, const u32 *__restrict__ start, u32 *cur,
			       u32 *out)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
	const bool ok = i < n;
	const u32 key = ok ? keys[i] : 0u;
	bool head;
	u32 len, pos;
	d_runs(key, ok, lane, head, len, pos);
	u32 b = 0;
	if (head)
		b = start[key] + atomicAdd(&cur[key], len);
	b = (u32)__shfl((int)b, (int)(lane - pos));
	if (ok)
		out[b + pos] = vals[i];
}

int main()
{
	const u32 n = 3000000u;
	for (int shape = 0; shape < 3; shape++) {
		for (u32 K : { 1024u, 17000u, 300000u }) {
			std::vector<u32> h(n), st(K + 1, 0u);
			u32 x = 12345u;
			for (u32 i = 0; i < n; i++) {
				x = x * 1664525u + 1013904223u;
				if (shape == 0)
					h[i] = (x >> 8) % K; // random
				else if (shape == 1)
					h[i] = ((i / 29u) * 2654435761u >> 8) % K; // runs of 29 equal keys
				else
					h[i] = ((i / 400u) * 7u + ((x >> 10) & 7u)) % K; // a cull flush: a few neighbouring buckets at a time
			}
			for (u32 i = 0; i < n; i++)
				st[h[i] + 1]++;
			for (u32 k = 0; k < K; k++)
				st[k + 1] += st[k];
			u32 *keys, *vals, *cnt, *start, *out;
			CHECK(hipMalloc(&keys, n * 4));
			CHECK(hipMalloc(&vals, n * 4));
			CHECK(hipMalloc(&out, n * 4));
			CHECK(hipMalloc(&cnt, K * 4));
			CHECK(hipMalloc(&start, (K + 1) * 4));
			CHECK(hipMemcpy(keys, h.data(), n * 4, hipMemcpyHostToDevice));
			CHECK(hipMemcpy(vals, h.data(), n * 4, hipMemcpyHostToDevice));
			CHECK(hipMemcpy(start, st.data(), (K + 1) * 4, hipMemcpyHostToDevice));
			hipEvent_t a, b;
			CHECK(hipEventCreate(&a));
			CHECK(hipEventCreate(&b));
			const dim3 grid((n + 255) / 256), block(256);
			float ms[4] = { 0, 0, 0, 0 };
			const int reps = 20;
			for (int v = 0; v < 4; v++) {
				for (int r = -2; r < reps; r++) {
					CHECK(hipMemsetAsync(cnt, 0, K * 4, 0));
					CHECK(hipEventRecord(a, 0));
					if (v == 0)
						hipLaunchKernelGGL(k_count_naive, grid, block, 0, 0, keys, n, cnt);
					else if (v == 1)
						hipLaunchKernelGGL(k_count_runs, grid, block, 0, 0, keys, n, cnt);
					else if (v == 2)
						hipLaunchKernelGGL(k_scatter_naive, grid, block, 0, 0, keys, vals, n, start, cnt, out);
					else
						hipLaunchKernelGGL(k_scatter_runs, grid, block, 0, 0, keys, vals, n, start, cnt, out);
					CHECK(hipEventRecord(b, 0));
					CHECK(hipEventSynchronize(b));
					float t;
					CHECK(hipEventElapsedTime(&t, a, b));
					if (r >= 0)
						ms[v] += t / reps;
				}
			}
			// the scatter fills every slot exactly once: checksum of the values against the input's
			std::vector<u32> o(n);
			CHECK(hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost));
			unsigned long long s0 = 0, s1 = 0;
			for (u32 i = 0; i < n; i++) {
				s0 += h[i];
				s1 += o[i];
			}
			printf("{\"pairs\": %u, \"buckets\": %u, \"keys\": \"%s\", \"count_naive_us\": %.1f, \"count_runs_us\": %.1f, "
			       "\"scatter_naive_us\": %.1f, \"scatter_runs_us\": %.1f, \"checksum_ok\": %s}\n",
			       n, K, shape == 0 ? "random" : (shape == 1 ? "runs of 29" : "flushes"), ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3,
			       ms[3] * 1e3, s0 == s1 ? "true" : "false");
			fflush(stdout);
			CHECK(hipFree(keys));
			CHECK(hipFree(vals));
			CHECK(hipFree(out));
			CHECK(hipFree(cnt));
			CHECK(hipFree(start));
		}
	}
	return 0;
}
