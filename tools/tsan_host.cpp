/* tools/tsan_host.cpp -- the host logic that several threads share, under ThreadSanitizer (`make -C pfac_amd/csrc tsan`):
 * one host-only handle used by N threads at once (PFAC_matchFromHost on the CPU platforms, PFACX_getInfo, PFACX_getTable,
 * PFAC_dumpTransitionTable, a PFAC_setPerfMode now and then), the way user code shares a handle (PFAC/README:135-137; the
 * reference serialises such threads with its texture mutex, PFAC.cpp:37-56).  CPU only.   tsan_host <threads> <rounds> <pattern file> */
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "PFAC.h"
#include "pfac_ext.h"

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 50;
    const char *patFile = argc > 3 ? argv[3] : nullptr;
    if (!patFile) { fprintf(stderr, "usage: tsan_host <threads> <rounds> <pattern file>\n"); return 2; }
    PFAC_handle_t h = nullptr;
    if (PFACX_createHostOnly(&h) != PFAC_STATUS_SUCCESS) return 2;
    if (PFAC_setPlatform(h, PFAC_PLATFORM_CPU) != PFAC_STATUS_SUCCESS) return 2;
    if (PFAC_readPatternFromFile(h, const_cast<char *>(patFile)) != PFAC_STATUS_SUCCESS) { fprintf(stderr, "pattern file refused\n"); return 2; }
    std::string input = "ABEDEDABG she sells hers his ABGABG";
    for (int i = 0; i < 10; i++) input += input;
    std::vector<int> want(input.size());
    if (PFAC_matchFromHost(h, const_cast<char *>(input.data()), input.size(), want.data()) != PFAC_STATUS_SUCCESS) return 2;
    std::atomic<int> bad{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t]() {
            std::vector<int> got(input.size());
            for (int r = 0; r < rounds; r++) {
                if (PFAC_matchFromHost(h, const_cast<char *>(input.data()), input.size(), got.data()) != PFAC_STATUS_SUCCESS || got != want) bad++;
                PFACX_info_t info;
                memset(&info, 0, sizeof(info));
                info.structSize = sizeof(info);
                if (PFACX_getInfo(h, &info) != PFAC_STATUS_SUCCESS) bad++;
                const void *p = nullptr;
                size_t bytes = 0;
                (void)PFACX_getTable(h, PFACX_TABLE_INITIAL_ROW, &p, &bytes);
                if (t == 0 && r % 10 == 5) (void)PFAC_setPerfMode(h, (r / 10) % 2 ? PFAC_SPACE_DRIVEN : PFAC_TIME_DRIVEN);   /* rebuilds the tables under the handle's lock */
            }
        });
    for (auto &th : pool) th.join();
    (void)PFAC_destroy(h);
    printf("tsan_host: %d threads x %d rounds on one handle: %d wrong results, no sanitizer report\n", threads, rounds, bad.load());
    return bad.load() ? 1 : 0;
}
