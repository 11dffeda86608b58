// How fast can a host Vec<Blob> (1 024 x 128 KiB, pageable) reach the GPU in S slices ACROSS blobs (slice j = bytes
// [j W, (j+1) W) of every blob, W = 128 KiB / S), so that the SHA-256 chains can consume slice j while slice j+1 is on the link?
//   A. hipMemcpy2DAsync from pageable memory, one call per slice          (does the runtime pin per call? staging?)
//   B. hipHostRegister once (cost on resident pages, repeated), then S async 2D copies from the registered range
//   C. T host threads gather slice j into a pinned buffer, one 1D DMA per slice (slice-major device layout)
//   hipcc -O3 --offload-arch=gfx950 -pthread -o hostslicebench hostslicebench.hip && ./hostslicebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 1024, BLOB = 131072, N = n * BLOB;
    char* d;
    CK(hipMalloc((void**)&d, N));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    char* host = (char*)malloc(N + 4096);
    memset(host, 1, N + 4096);
    char* src = host + 64;
    hipEvent_t ev[17];
    for (auto& evt : ev) CK(hipEventCreate(&evt));
    // baseline
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        CK(hipMemcpyAsync(d, src, N, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        printf("baseline 1D pageable: %.3f ms\n", (now() - t0) * 1e3);
    }
    // A
    for (int S : {2, 4, 8, 16}) {
        const size_t W = BLOB / S;
        for (int rep = 0; rep < 3; rep++) {
            double t0 = now(), first = 0, issue = 0;
            for (int j = 0; j < S; j++) {
                CK(hipMemcpy2DAsync(d + j * W, BLOB, src + j * W, BLOB, W, n, hipMemcpyHostToDevice, st));
                if (j == 0) issue = now() - t0;
            }
            double t_issue = now() - t0;
            CK(hipStreamSynchronize(st));
            (void)first;
            printf("A: pageable 2D, S = %2d: first call returns after %.3f ms, all issued after %.3f ms, done after %.3f ms\n", S, issue * 1e3, t_issue * 1e3, (now() - t0) * 1e3);
        }
    }
    // B: register cost on the SAME resident buffer, repeated; then async 2D slices with per-slice arrival times
    for (int rep = 0; rep < 6; rep++) {
        double t0 = now();
        CK(hipHostRegister(src, N, hipHostRegisterDefault));
        double t1 = now();
        const int S = 8;
        const size_t W = BLOB / S;
        CK(hipEventRecord(ev[16], st));
        for (int j = 0; j < S; j++) {
            CK(hipMemcpy2DAsync(d + j * W, BLOB, src + j * W, BLOB, W, n, hipMemcpyHostToDevice, st));
            CK(hipEventRecord(ev[j], st));
        }
        double t2 = now();
        CK(hipStreamSynchronize(st));
        double t3 = now();
        CK(hipHostUnregister(src));
        double t4 = now();
        float a0, a7;
        CK(hipEventElapsedTime(&a0, ev[16], ev[0]));
        CK(hipEventElapsedTime(&a7, ev[16], ev[7]));
        printf("B: register %.3f ms, 8 x 2D issue %.3f ms, copies done %.3f ms after issue start (slice 0 landed at %.3f, slice 7 at %.3f), unregister %.3f ms, total %.3f ms\n",
               (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t1) * 1e3, a0, a7, (t4 - t3) * 1e3, (t4 - t0) * 1e3);
    }
    // B2: register in P pieces on P threads
    for (int P : {2, 4, 8}) {
        for (int rep = 0; rep < 2; rep++) {
            double t0 = now();
            std::vector<std::thread> th;
            for (int p = 0; p < P; p++) th.emplace_back([&, p] { CK(hipHostRegister(src + (N / P) * p, N / P, hipHostRegisterDefault)); });
            for (auto& t : th) t.join();
            double t1 = now();
            for (int p = 0; p < P; p++) CK(hipHostUnregister(src + (N / P) * p));
            printf("B2: register in %d pieces on %d threads: %.3f ms (unregister %.3f ms)\n", P, P, (t1 - t0) * 1e3, (now() - t1) * 1e3);
        }
    }
    // B3: 1D async copies from a registered buffer in K row chunks (whole blobs), for comparison
    {
        CK(hipHostRegister(src, N, hipHostRegisterDefault));
        for (int rep = 0; rep < 2; rep++) {
            double t0 = now();
            for (int j = 0; j < 8; j++) CK(hipMemcpyAsync(d + j * (N / 8), src + j * (N / 8), N / 8, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            printf("B3: registered, 8 x 1D chunks: %.3f ms\n", (now() - t0) * 1e3);
        }
        CK(hipHostUnregister(src));
    }
    // C: host gather into pinned + 1D DMA per slice
    char* pinned;
    CK(hipHostMalloc((void**)&pinned, N, hipHostMallocDefault));
    memset(pinned, 2, N);
    for (int T : {4, 8, 16}) {
        const int S = 8;
        const size_t W = BLOB / S;
        for (int rep = 0; rep < 2; rep++) {
            double t0 = now();
            for (int j = 0; j < S; j++) {
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++)
                    th.emplace_back([&, t] {
                        for (size_t i = n * t / T; i < n * (t + 1) / T; i++) memcpy(pinned + j * (n * W) + i * W, src + i * BLOB + j * W, W);
                    });
                for (auto& t : th) t.join();
                CK(hipMemcpyAsync(d + j * (n * W), pinned + j * (n * W), n * W, hipMemcpyHostToDevice, st));
            }
            double t1 = now();
            CK(hipStreamSynchronize(st));
            printf("C: %2d threads gather + DMA per slice (S = 8): gathers done %.3f ms, all landed %.3f ms\n", T, (t1 - t0) * 1e3, (now() - t0) * 1e3);
        }
    }
    return 0;
}
