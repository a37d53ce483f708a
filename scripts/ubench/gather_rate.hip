// Micro-benchmark: rate at which a CU's vector L1 serves 16-byte gathers on gfx950, in the shapes the hierarchy walk uses.
// Every lane reads 64-byte records at pseudo-random indices of a table that is resident in L2 (or in L1: small table).
//   kind 0: one 16-byte read per record (every lane-read in a cache line of its own)
//   kind 1: the four 16-byte reads of a record from one lane, in four instructions (the walk's node fetch)
//   kind 2: the four lanes of a quad read the four 16-byte words of one record in one instruction, four instructions for
//           the quad's four records (same bytes as kind 1, a quarter of the distinct lines per instruction)
//   kind 5 / 6: one 4-byte / one 8-byte read per record (is the rate one of lane-reads or of bytes?)
//   kind 3 / 4: kind 1 with half of the lanes switched off: lanes 32-63 (eight full quads left) / every other lane (all
//           sixteen quads still touched) — does a read cost per lane or per quad of lanes?
// Occupancy is held to 16 one-wave workgroups per CU by LDS, as in the walk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int KIND>
__global__ __launch_bounds__(64) void k(const float4 *table, unsigned mask, int iters, float *out)
{
	extern __shared__ float pad[];
	unsigned s = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u;
	float acc = 0.0f;
	const int quad_lane = threadIdx.x & 3;
	for (int i = 0; i < iters; ++i)
	{
		s = s * 1664525u + 1013904223u;
		const unsigned idx = (s >> 8) & mask;
		if (KIND == 0)
		{
			const float4 a = table[4 * (size_t)idx];
			acc += a.x;
		}
		if (KIND == 5) acc += reinterpret_cast<const float *>(table)[16 * (size_t)idx];
		if (KIND == 6) { const float2 a = reinterpret_cast<const float2 *>(table)[8 * (size_t)idx]; acc += a.x + a.y; }
		if (KIND == 3 && threadIdx.x >= 32) continue;
		if (KIND == 4 && (threadIdx.x & 1)) continue;
		if (KIND == 1 || KIND == 3 || KIND == 4)
		{
			const float4 a = table[4 * (size_t)idx], b = table[4 * (size_t)idx + 1], c = table[4 * (size_t)idx + 2], d = table[4 * (size_t)idx + 3];
			acc += a.x + b.y + c.z + d.w;
		}
		if (KIND == 2)
		{
			const unsigned i0 = __shfl(idx, 0, 4), i1 = __shfl(idx, 1, 4), i2 = __shfl(idx, 2, 4), i3 = __shfl(idx, 3, 4);
			const float4 a = table[4 * (size_t)i0 + quad_lane], b = table[4 * (size_t)i1 + quad_lane];
			const float4 c = table[4 * (size_t)i2 + quad_lane], d = table[4 * (size_t)i3 + quad_lane];
			acc += a.x + b.y + c.z + d.w;
		}
	}
	if (pad[threadIdx.x] != 0.0f) acc += 1.0f;
	out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <int KIND> void run(const char *name, const float4 *table, unsigned records, int reads, float *out)
{
	const int iters = 4000, grid = 256 * 16 * 4;
	const size_t lds = 9728;
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), lds, 0, table, records - 1, 10, out);
	hipEventRecord(a);
	hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), lds, 0, table, records - 1, iters, out);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	const double lane_reads = (double)grid * 64 * iters * reads, cycles = ms * 1e-3 * 2.4e9 * 256; // CU-cycles at 2.4 GHz
	printf("%-46s table %6u KB  %7.2f ms  %5.2f lane-reads (16 B) per cycle and CU  %6.2f TB/s\n", name, records * 64 / 1024, ms,
	       lane_reads / cycles, lane_reads * 16 / (ms * 1e-3) / 1e12);
}
int main()
{
	const unsigned big = 1u << 15, small = 1u << 7; // 2 MB (L2), 8 KB (L1)
	float4 *table; float *out;
	hipMalloc(&table, (size_t)big * 64); hipMemset(table, 0, (size_t)big * 64);
	hipMalloc(&out, 256 * 16 * 4 * 64 * sizeof(float));
	for (unsigned n : { big, small })
	{
		run<0>("one read per record", table, n, 1, out);
		run<1>("four reads of a record from one lane", table, n, 4, out);
		run<2>("a record's four words from the four lanes of a quad", table, n, 4, out);
		run<5>("one 4-byte read per record (lane-reads of 4 B)", table, n, 1, out);
		run<6>("one 8-byte read per record (lane-reads of 8 B)", table, n, 1, out);
		run<3>("four reads from one lane, lanes 0-31 only (x2 = per active lane)", table, n, 2, out);
		run<4>("four reads from one lane, even lanes only (x2 = per active lane)", table, n, 2, out);
	}
	return 0;
}
