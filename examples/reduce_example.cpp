// reduce_example.cpp -- the reference's compacted-output example (PFAC/test/simple_example_reduce.cpp; known answer in
// the user guide r1.2 p.29) against this repo's drop-in library: PFAC_matchFromHostReduce in both perf modes, and
// PFAC_matchFromDeviceReduce with caller-managed HIP buffers; prints "number of matched = K" and one line per match.
//
//   make -C examples && ./examples/reduce_example [pattern_file input_file]
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "PFAC.h"

static void check(const char *what, PFAC_status_t st)
{
    if (st == PFAC_STATUS_SUCCESS) return;
    std::fprintf(stderr, "%s: %s\n", what, PFAC_getErrorString(st));
    std::exit(1);
}

int main(int argc, char **argv)
{
    const char *patternFile = argc > 2 ? argv[1] : "tests/golden/example_pattern";
    const char *inputFile = argc > 2 ? argv[2] : "tests/golden/example_input";
    std::FILE *fp = std::fopen(inputFile, "rb");
    if (!fp) { std::perror(inputFile); return 1; }
    std::fseek(fp, 0, SEEK_END);
    const size_t n = (size_t)std::ftell(fp);
    std::rewind(fp);
    std::vector<char> input(n);
    if (std::fread(input.data(), 1, n, fp) != n) return 1;
    std::fclose(fp);

    PFAC_handle_t handle;
    check("PFAC_create", PFAC_create(&handle));
    check("PFAC_readPatternFromFile", PFAC_readPatternFromFile(handle, const_cast<char *>(patternFile)));

    std::vector<int> id[3], pos[3];
    int count[3] = {0, 0, 0};
    const PFAC_perfMode_t modes[2] = {PFAC_TIME_DRIVEN, PFAC_SPACE_DRIVEN};
    for (int m = 0; m < 2; m++) {
        check("PFAC_setPerfMode", PFAC_setPerfMode(handle, modes[m]));
        id[m].assign(n, 0); pos[m].assign(n, 0);
        check("PFAC_matchFromHostReduce", PFAC_matchFromHostReduce(handle, input.data(), n, id[m].data(), pos[m].data(), &count[m]));
    }
    char *d_in = nullptr;
    int *d_id = nullptr, *d_pos = nullptr;
    if (hipMalloc((void **)&d_in, (n + 3) / 4 * 4) != hipSuccess || hipMalloc((void **)&d_id, n * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&d_pos, n * sizeof(int)) != hipSuccess) return 1;
    (void)hipMemcpy(d_in, input.data(), n, hipMemcpyHostToDevice);
    check("PFAC_matchFromDeviceReduce", PFAC_matchFromDeviceReduce(handle, d_in, n, d_id, d_pos, &count[2]));
    id[2].assign(n, 0); pos[2].assign(n, 0);
    (void)hipMemcpy(id[2].data(), d_id, (size_t)count[2] * sizeof(int), hipMemcpyDeviceToHost);
    (void)hipMemcpy(pos[2].data(), d_pos, (size_t)count[2] * sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d_in); (void)hipFree(d_id); (void)hipFree(d_pos);

    std::printf("number of matched = %d\n", count[0]);
    for (int i = 0; i < count[0]; i++) std::printf("At position %4d, match pattern %d\n", pos[0][i], id[0][i]);
    int bad = 0;
    for (int v = 1; v < 3; v++) {
        if (count[v] != count[0]) bad++;
        for (int i = 0; i < count[0] && i < count[v]; i++) bad += (pos[v][i] != pos[0][i]) || (id[v][i] != id[0][i]);
    }
    check("PFAC_destroy", PFAC_destroy(handle));
    if (bad) { std::fprintf(stderr, "the three compacted results disagree (%d differences)\n", bad); return 2; }
    return 0;
}
