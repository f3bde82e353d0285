#include "rendezvous.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace mkhost {

namespace {

const char kMagic[8] = {'M', 'K', 'C', 'O', 'M', 'M', '2', '\n'};

// field 22 of /proc/<pid>/stat: the process's start time in clock ticks since boot (0: no such process / no /proc)
unsigned long long start_time_of(long pid)
{
    char path[64];
    snprintf(path, sizeof path, "/proc/%ld/stat", pid);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    char buf[2048];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *p = strrchr(buf, ')');                              // (the command name may hold spaces and parentheses)
    if (!p) return 0;
    ++p;
    unsigned long long v = 0;
    for (int field = 3; field <= 22; ++field) {
        while (*p == ' ') ++p;
        if (!*p) return 0;
        if (field == 22) { v = strtoull(p, nullptr, 10); break; }
        while (*p && *p != ' ') ++p;
    }
    return v;
}

uint64_t fnv1a(const uint8_t *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

void put64(std::vector<uint8_t> &b, uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
uint64_t get64(const uint8_t *p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i); return v; }

std::string safe(const char *s)
{
    std::string o;
    for (; s && *s && o.size() < 48; ++s) o += (isalnum((unsigned char)*s) || *s == '-' || *s == '_') ? *s : '_';
    return o;
}

}  // namespace

Rendezvous rendezvous_from_env()
{
    Rendezvous r;
    if (const char *e = getenv("MIEKKI_COMM_NONCE")) r.nonce = e;
    else {
        const long pp = (long)getppid();
        r.nonce = std::to_string((unsigned long)geteuid()) + "." + std::to_string(pp) + "." + std::to_string(start_time_of(pp)) + "." +
                  safe(getenv("MASTER_PORT")) + "." + safe(getenv("TORCHELASTIC_RUN_ID"));
    }
    if (const char *e = getenv("MIEKKI_COMM_FILE")) { r.path = e; return r; }
    const char *dir = getenv("XDG_RUNTIME_DIR");
    if (!dir || !*dir) dir = getenv("TMPDIR");
    if (!dir || !*dir) dir = "/tmp";
    r.path = std::string(dir) + "/miekki_comm_" + safe(r.nonce.c_str());
    return r;
}

bool rendezvous_publish(const Rendezvous &r, const void *payload, size_t n, std::string &err)
{
    std::vector<uint8_t> b(kMagic, kMagic + 8);
    put64(b, r.nonce.size());
    b.insert(b.end(), r.nonce.begin(), r.nonce.end());
    put64(b, (uint64_t)getpid());
    put64(b, start_time_of((long)getpid()));
    put64(b, n);
    b.insert(b.end(), (const uint8_t *)payload, (const uint8_t *)payload + n);
    put64(b, fnv1a(b.data(), b.size()));
    (void)unlink(r.path.c_str());                                   // a file left behind (or planted) under this name: gone
    const std::string tmp = r.path + ".tmp." + std::to_string((long)getpid());
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) { err = "cannot create " + tmp + ": " + strerror(errno); return false; }
    size_t at = 0;
    while (at < b.size()) {
        const ssize_t w = write(fd, b.data() + at, b.size() - at);
        if (w <= 0) { err = "cannot write " + tmp + ": " + strerror(errno); close(fd); unlink(tmp.c_str()); return false; }
        at += (size_t)w;
    }
    if (close(fd) != 0 || rename(tmp.c_str(), r.path.c_str()) != 0) { err = "cannot write " + r.path + ": " + strerror(errno); unlink(tmp.c_str()); return false; }
    return true;
}

bool rendezvous_fetch(const Rendezvous &r, void *payload, size_t n, long timeout_ms, std::string &err)
{
    const bool pid_check = !getenv("MIEKKI_COMM_NO_PID_CHECK");
    const auto t0 = std::chrono::steady_clock::now();
    std::string why = "no such file";
    for (;;) {
        const int fd = open(r.path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
        if (fd >= 0) {
            struct stat st;
            std::vector<uint8_t> b;
            if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && st.st_size >= 48 && st.st_size < (1 << 20)) {
                b.resize((size_t)st.st_size);
                size_t at = 0;
                while (at < b.size()) { const ssize_t g = read(fd, b.data() + at, b.size() - at); if (g <= 0) break; at += (size_t)g; }
                if (at != b.size()) b.clear();
            } else why = "not a regular file of this user";
            close(fd);
            // magic | nonce length, nonce | pid | start | n, payload | checksum
            if (b.size() >= 48 && !memcmp(b.data(), kMagic, 8) && get64(b.data() + b.size() - 8) == fnv1a(b.data(), b.size() - 8)) {
                const uint64_t nl = get64(b.data() + 8);
                if (nl < b.size() && 16 + nl + 24 + 8 <= b.size()) {
                    const uint8_t *p = b.data() + 16 + nl;
                    const uint64_t pid = get64(p), start = get64(p + 8), pn = get64(p + 16);
                    const bool mine = std::string((const char *)b.data() + 16, (size_t)nl) == r.nonce;
                    const bool alive = !pid_check || (start_time_of((long)pid) == start && start != 0);
                    if (!mine) why = "another run's file (nonce)";
                    else if (!alive) why = "left behind by a run whose rank 0 is gone (ranks in separate pid namespaces, or a /proc that hides other users' "
                                           "processes, cannot see rank 0: set MIEKKI_COMM_NO_PID_CHECK=1)";
                    else if (pn != n || 16 + nl + 24 + pn + 8 != b.size()) why = "payload of another size";
                    else { memcpy(payload, p + 24, n); return true; }
                } else why = "malformed";
            } else if (!b.empty()) why = "not a rendezvous file (magic / checksum)";
        }
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > (double)timeout_ms) {
            err = "no communicator id in " + r.path + " (" + why + ")";
            return false;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
}

void rendezvous_remove(const Rendezvous &r)
{
    if (!r.path.empty()) (void)unlink(r.path.c_str());
}

}  // namespace mkhost
