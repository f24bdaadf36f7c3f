// cda_tools.hip - measuring probes, NOT part of the product library (libcda_hip.so): built into tools/libcda_tools.so by
// __graft_entry__.build() and loaded only by tools/opbench.py, tools/clock_probe.py and tools/pmc_calib.py.
// They share the product's device headers (the decimal arithmetic under test) and nothing else.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cda.h"
#include "../../gym_continuousdoubleauction_amd/csrc/cda_dec.hpp"
#include "../../gym_continuousdoubleauction_amd/csrc/cda_market.hpp"

using namespace cda;

// debug micro-benchmark (tools/opbench.py): cycles of one decimal operation on representative ledger operands,
// measured on a single wave (one active lane) as a dependent chain of `iters` operations
__global__ void k_opbench(int op, int iters, const cda_dec* a, const cda_dec* b, unsigned long long* out, cda_dec* sink) {
    dec_tables_init();
    if (threadIdx.x != 0) return;
    D x = ld_dec(a[0]), y = ld_dec(b[0]);
    uint32_t f = 0; double dacc = 0.0; int iacc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        switch (op) {
            case 0: x = d_add(x, y); y.sign ^= 1; break;                       // add / sub alternating: value stays bounded
            case 1: { D r = d_mul_u32(x, y.w0, 0); x.w0 = (x.w0 ^ r.w0) | 1u; break; }
            case 2: { D r = d_div_u32(x, y.w0); x.w0 = (x.w0 ^ r.w0) | 1u; break; }
            case 3: iacc += d_cmp(x, y); x.w0 ^= (uint32_t)iacc; break;
            case 4: dacc += d_to_double(x, &f); x.w0 ^= (uint32_t)__double_as_longlong(dacc); break;
            default: break;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[0] = t1 - t0;
    st_dec(sink[0], x, f); sink[1].w[0] = (uint32_t)iacc + (uint32_t)__double_as_longlong(dacc);
}

// debug (tools/clock_probe.py): shader-clock cycles (s_memtime) against the constant 100 MHz counter over a spin
__global__ void k_clock_probe(int iters, unsigned long long* out) {
    unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    unsigned int x = threadIdx.x;
    for (int i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
    unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = x; }
}
// PMC calibration (tools/profile_gpu.sh): known byte counts in THIS library's access pattern (4 B per lane,
// coalesced) so that FETCH_SIZE / WRITE_SIZE can be converted to bytes (MI355X_MICROARCH.md, HBM section).
__global__ void k_calib_read(const uint32_t* p, size_t n, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += stride) acc ^= p[i];
    if (acc == 0x12345679u) out[0] = acc;
}
__global__ void k_calib_write(uint32_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (uint32_t)i;
}


extern "C" {

/* debug hook (not in include/cda.h): cycles of `iters` dependent decimal operations of kind `op` (see k_opbench) */
long long cda_debug_opbench(int op, int iters, const cda_dec* a_host, const cda_dec* b_host, const void* acc_host /* 144 B or NULL */) {
    cda_dec *da = NULL, *db = NULL, *ds = NULL; unsigned long long* dout = NULL; unsigned long long cyc = 0;
    if (hipMalloc((void**)&da, 16) != hipSuccess || hipMalloc((void**)&db, 16) != hipSuccess || hipMalloc((void**)&ds, 32) != hipSuccess ||
        hipMalloc((void**)&dout, 8) != hipSuccess) return -1;
    (void)hipMemcpy(da, a_host, 16, hipMemcpyHostToDevice); (void)hipMemcpy(db, b_host, 16, hipMemcpyHostToDevice);
    (void)acc_host;
    hipLaunchKernelGGL(k_opbench, dim3(1), dim3(64), DEC_TABLE_BYTES + 256, 0, op, iters, da, db, dout, ds);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    (void)hipMemcpy(&cyc, dout, 8, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(ds); (void)hipFree(dout);
    return (long long)cyc;
}

/* debug hook (not in include/cda.h): read (mode 0) or write (mode 1) n_bytes of a device buffer, 4 B per lane */
int cda_debug_calib(void* dev_buf, size_t n_bytes, int mode, void* stream) {
    size_t n = n_bytes / 4;
    if (mode == 0) hipLaunchKernelGGL(k_calib_read, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)dev_buf, n, (uint32_t*)dev_buf);
    else hipLaunchKernelGGL(k_calib_write, dim3(2048), dim3(256), 0, (hipStream_t)stream, (uint32_t*)dev_buf, n);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

/* debug hook (not in include/cda.h): enqueue the clock probe on `stream`; out: device u64[3] = shader cycles, 100 MHz ticks, sink */
int cda_debug_clock_probe(int iters, unsigned long long* dev_out, void* stream) {
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, iters, dev_out);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

}  // extern "C"
