/* tools/fuzz_host.cpp -- mutation loop over the host code that parses untrusted bytes: PFACX_loadCompiled and the four
 * pattern readers (PFAC_readPatternFromFile, PFACX_readPatternFromMemory, ...FromFileEx, ...FromMemoryEx).  Linked against the
 * AddressSanitizer + UBSan build of libpfac.so (`make -C pfac_amd/csrc san`); CPU only (host-only handles).
 *
 *   fuzz_host <iterations> <seed> <scratch dir>
 *
 * Every iteration takes a valid compiled set / pattern file, changes 1..8 bytes (or cuts / extends it), RECOMPUTES the FNV-1a
 * checksum of a compiled set (a checksum anyone can recompute protects nothing), loads it, and -- if the library accepts
 * it -- matches a small input on the CPU platform, so that tables built from accepted bytes are walked too.  Any status is
 * fine; a sanitizer report or a crash is the failure.  Prints a summary line (seed, iterations, accepted / refused). */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "PFAC.h"
#include "pfac_ext.h"

static uint64_t rngState;
static uint64_t rnd()
{
    uint64_t z = (rngState += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static uint64_t fnv1a(const unsigned char *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}
static void writeFile(const std::string &path, const std::vector<unsigned char> &bytes)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror("fopen"); exit(2); }
    if (!bytes.empty()) fwrite(bytes.data(), 1, bytes.size(), f);
    fclose(f);
}
static std::vector<unsigned char> readFile(const std::string &path)
{
    std::vector<unsigned char> v;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return v;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}
static void mutate(std::vector<unsigned char> &b)
{
    const unsigned kind = (unsigned)(rnd() % 10);
    if (b.empty()) { b.push_back((unsigned char)rnd()); return; }
    if (kind == 0 && b.size() > 1) { b.resize((size_t)(rnd() % b.size())); return; }                    /* cut */
    if (kind == 1) { const size_t extra = (size_t)(rnd() % 64); for (size_t i = 0; i < extra; i++) b.push_back((unsigned char)rnd()); return; }
    if (kind == 2) {                                                                                     /* a 32-bit field set to an extreme */
        static const uint32_t extremes[] = {0u, 1u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFFu, 0xFFFFu, 0x10000u};
        if (b.size() >= 4) { const size_t at = (size_t)(rnd() % (b.size() - 3)) & ~size_t(3); const uint32_t v = extremes[rnd() % 7]; memcpy(&b[at], &v, 4); }
        return;
    }
    const unsigned flips = 1 + (unsigned)(rnd() % 8);
    for (unsigned k = 0; k < flips; k++) b[(size_t)(rnd() % b.size())] = (unsigned char)rnd();
}

int main(int argc, char **argv)
{
    const long iterations = argc > 1 ? atol(argv[1]) : 1000;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1;
    const std::string dir = argc > 3 ? argv[3] : "/tmp";
    rngState = seed;
    /* a pattern set with everything the compiler branches on: short and long patterns, prefixes of patterns, bytes >= 0x80,
     * a long shared prefix, CRLF-looking bytes */
    std::string pats;
    const char *fixed[] = {"AB", "ABG", "BEDE", "ED", "a", "he", "she", "hers", "his", "GET /index.html HTTP/1.1", "User-Agent: Mozilla/5.0",
                           "User-Agent: curl/8.4.0", "abcdefghijklmnopqrstuvwxyz0123456789abcdefghijklmnopqrstuvwxyz", "\x80\xff\x01", "x\ry"};
    for (const char *p : fixed) { pats += p; pats += "\n"; }
    for (int i = 0; i < 40; i++) {
        std::string p = "common-prefix-of-24-bytes";
        const int tail = 3 + (int)(rnd() % 30);
        for (int k = 0; k < tail; k++) p += (char)('a' + rnd() % 26);
        pats += p + "\n";
    }
    const std::string patFile = dir + "/fuzz_host.pat", setFile = dir + "/fuzz_host.pfacx", mutFile = dir + "/fuzz_host.mut";
    writeFile(patFile, std::vector<unsigned char>(pats.begin(), pats.end()));
    PFAC_handle_t good = nullptr, h = nullptr;
    if (PFACX_createHostOnly(&good) != PFAC_STATUS_SUCCESS || PFACX_createHostOnly(&h) != PFAC_STATUS_SUCCESS) { fprintf(stderr, "createHostOnly failed\n"); return 2; }
    if (PFAC_readPatternFromFile(good, const_cast<char *>(patFile.c_str())) != PFAC_STATUS_SUCCESS) { fprintf(stderr, "valid pattern file refused\n"); return 2; }
    if (PFACX_saveCompiled(good, setFile.c_str()) != PFAC_STATUS_SUCCESS) { fprintf(stderr, "saveCompiled failed\n"); return 2; }
    const std::vector<unsigned char> set = readFile(setFile);
    if (set.size() < 40) { fprintf(stderr, "compiled set too small\n"); return 2; }
    std::string input = "xxUser-Agent: curl/8.4.0 ABEDEDABG she hers common-prefix-of-24-bytesabcdefghij GET /index.html HTTP/1.1\n";
    for (int i = 0; i < 8; i++) input += input.substr(0, 64);
    std::vector<int> result(input.size());
    (void)PFAC_setPlatform(h, PFAC_PLATFORM_CPU);
    long accepted = 0, refused = 0;
    for (long it = 0; it < iterations; it++) {
        const unsigned which = (unsigned)(rnd() % 6);
        PFAC_status_t st;
        if (which < 2) {                                         /* a compiled set; checksum recomputed in two cases out of three */
            std::vector<unsigned char> b = set;
            mutate(b);
            if (b.size() >= 40 && rnd() % 3 != 0) {
                uint64_t payload = b.size() - 40;
                if (rnd() % 4 != 0) memcpy(&b[24], &payload, 8);   /* payloadBytes */
                const uint64_t sum = fnv1a(b.data() + 40, b.size() - 40);
                memcpy(&b[32], &sum, 8);
            }
            writeFile(mutFile, b);
            st = PFACX_loadCompiled(h, mutFile.c_str());
        } else {                                                 /* pattern text through one of the four readers */
            std::vector<unsigned char> b(pats.begin(), pats.end());
            mutate(b);
            if (rnd() % 5 == 0) { const size_t at = b.empty() ? 0 : (size_t)(rnd() % b.size()); b.insert(b.begin() + at, (unsigned char)'\n'); b.insert(b.begin() + at, (unsigned char)'\r'); }
            const unsigned flags = (unsigned)(rnd() % 4);
            if (which == 2) { writeFile(mutFile, b); st = PFAC_readPatternFromFile(h, const_cast<char *>(mutFile.c_str())); }
            else if (which == 3) { writeFile(mutFile, b); st = PFACX_readPatternFromFileEx(h, mutFile.c_str(), flags); }
            else if (which == 4) st = PFACX_readPatternFromMemory(h, reinterpret_cast<const char *>(b.data()), b.size());
            else st = PFACX_readPatternFromMemoryEx(h, reinterpret_cast<const char *>(b.data()), b.size(), flags);
        }
        if (st == PFAC_STATUS_SUCCESS) {
            accepted++;
            (void)PFAC_setPerfMode(h, (it & 1) ? PFAC_SPACE_DRIVEN : PFAC_TIME_DRIVEN);
            if (PFAC_matchFromHost(h, const_cast<char *>(input.data()), input.size(), result.data()) != PFAC_STATUS_SUCCESS) { fprintf(stderr, "match on an accepted set failed (iteration %ld)\n", it); return 1; }
            PFACX_info_t info;
            memset(&info, 0, sizeof(info));
            info.structSize = sizeof(info);
            (void)PFACX_getInfo(h, &info);
        } else {
            refused++;
        }
    }
    (void)PFAC_destroy(h);
    (void)PFAC_destroy(good);
    printf("fuzz_host: seed %llu, %ld iterations over PFACX_loadCompiled + 4 pattern readers: %ld accepted (and matched), %ld refused, no sanitizer report\n",
           (unsigned long long)seed, iterations, accepted, refused);
    return 0;
}
