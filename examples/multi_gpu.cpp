/*
 * multi_gpu.cpp -- one call shards a host stream over the GPUs of the node.
 *
 * What every user of the reference writes by hand after PFAC/test/omp_PFAC.cpp:257-394 (one OpenMP thread per
 * GPU, static chunks with a max_patternLen + 1 tail, results of the tail discarded) is a library call here:
 * PFACX_matchFromHostMultiGPU (include/pfac_ext.h).  Like omp_PFAC.cpp:396-439 the program checks the sharded
 * result element by element against a single-device scan.
 *
 *     ./multi_gpu <pattern file> <input file> [workers]      (workers > visible GPUs: devices are reused)
 */
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "PFAC.h"
#include "pfac_ext.h"

#define CHECK(call)                                                                             \
    do {                                                                                        \
        PFAC_status_t st_ = (call);                                                             \
        if (st_ != PFAC_STATUS_SUCCESS) {                                                       \
            std::fprintf(stderr, "%s failed: %s\n", #call, PFAC_getErrorString(st_));           \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s <pattern file> <input file> [workers]\n", argv[0]); return 2; }
    FILE *fp = std::fopen(argv[2], "rb");
    if (!fp) { std::fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
    std::fseek(fp, 0, SEEK_END);
    const size_t n = (size_t)std::ftell(fp);
    std::rewind(fp);
    std::vector<char> input(n);
    if (n && std::fread(input.data(), 1, n, fp) != n) { std::fclose(fp); return 2; }
    std::fclose(fp);

    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) { std::fprintf(stderr, "no GPU\n"); return 3; }
    const int workers = argc > 3 ? std::atoi(argv[3]) : visible;
    std::vector<int> devices;
    for (int i = 0; i < workers; i++) devices.push_back(i % visible);

    PFAC_handle_t handle;
    CHECK(PFAC_create(&handle));
    CHECK(PFAC_setPerfMode(handle, PFAC_SPACE_DRIVEN));
    CHECK(PFAC_readPatternFromFile(handle, argv[1]));

    std::vector<int> sharded(n, -1), single(n, -1);
    CHECK(PFACX_matchFromHostMultiGPU(handle, input.data(), n, sharded.data(), workers, devices.data()));
    CHECK(PFAC_matchFromHost(handle, input.data(), n, single.data()));

    size_t matches = 0, differ = 0;
    for (size_t i = 0; i < n; i++) {
        matches += single[i] != 0;
        differ += single[i] != sharded[i];
    }
    std::printf("%zu bytes, %d worker(s) on %d device(s): %zu matches, %zu differences between the sharded and the single scan\n",
                n, workers, visible, matches, differ);

    /* the compacted-output form: only the (id, position) pairs of the matches come back, in position order (one byte per position
     * over a host link, nothing filled on the host: what scales with the number of links) */
    std::vector<int> ids(n), pos(n);
    int pairs = 0;
    CHECK(PFACX_matchFromHostReduceMultiGPU(handle, input.data(), n, ids.data(), pos.data(), &pairs, workers, devices.data()));
    size_t wrong = (size_t)pairs != matches;
    for (int k = 0; k < pairs && !wrong; k++)
        wrong += pos[k] < 0 || (size_t)pos[k] >= n || single[pos[k]] != ids[k] || (k && pos[k] <= pos[k - 1]);
    std::printf("compacted output over the same workers: %d pairs, %s\n", pairs, wrong ? "DIFFERENT from the full result" : "equal to the non-zero entries of the full result");
    CHECK(PFAC_destroy(handle));
    return differ || wrong ? 1 : 0;
}
