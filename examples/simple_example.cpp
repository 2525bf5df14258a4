// simple_example.cpp -- the reference's first example (PFAC/test/simple_example.cpp, README.md:29-120)
// re-authored against this repo's drop-in library: compile patterns, scan a small input through
// PFAC_matchFromHost (GPU platform) and through PFAC_matchFromDevice with caller-managed HIP buffers,
// print "At position N, match pattern K" lines.
//
//   make -C examples && ./examples/simple_example [pattern_file input_file]
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "PFAC.h"

static void die(const char *what, PFAC_status_t st)
{
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
    PFAC_status_t st = PFAC_create(&handle);
    if (st != PFAC_STATUS_SUCCESS) die("PFAC_create", st);
    st = PFAC_readPatternFromFile(handle, const_cast<char *>(patternFile));
    if (st != PFAC_STATUS_SUCCESS) die("PFAC_readPatternFromFile", st);
    PFAC_dumpTransitionTable(handle, stdout);

    // (1) host buffers: the library stages them through the device
    std::vector<int> viaHost(n, -1);
    st = PFAC_matchFromHost(handle, input.data(), n, viaHost.data());
    if (st != PFAC_STATUS_SUCCESS) die("PFAC_matchFromHost", st);

    // (2) device buffers owned by the caller (README example 2)
    char *d_in = nullptr;
    int *d_out = nullptr;
    if (hipMalloc((void **)&d_in, (n + 3) / 4 * 4) != hipSuccess || hipMalloc((void **)&d_out, n * sizeof(int)) != hipSuccess) return 1;
    (void)hipMemcpy(d_in, input.data(), n, hipMemcpyHostToDevice);
    st = PFAC_matchFromDevice(handle, d_in, n, d_out);
    if (st != PFAC_STATUS_SUCCESS) die("PFAC_matchFromDevice", st);
    std::vector<int> viaDevice(n, -1);
    (void)hipMemcpy(viaDevice.data(), d_out, n * sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);

    int bad = 0;
    for (size_t i = 0; i < n; i++) {
        if (viaHost[i] != viaDevice[i]) bad++;
        if (viaHost[i] != 0) std::printf("At position %4zu, match pattern %d\n", i, viaHost[i]);
    }
    PFAC_destroy(handle);
    if (bad) { std::fprintf(stderr, "matchFromHost and matchFromDevice disagree at %d positions\n", bad); return 2; }
    return 0;
}
