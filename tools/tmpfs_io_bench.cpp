// How a memory file system takes one large file from 1 .. N threads (pwrite at disjoint offsets, or copies into a mapping) and gives
// it back (pread, or copies out of a mapping): what decided the index writer's single output thread (host/gzpar.cpp).
//   g++ -O2 -o tmpfs_io_bench tools/tmpfs_io_bench.cpp -lpthread && ./tmpfs_io_bench <0 = pwrite/pread | 1 = mmap | 2 = one thread
//   fallocates the file's pages 1 GiB ahead of <threads> threads that pwrite into them> <threads> <GiB>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <sched.h>
#include <thread>
#include <vector>
#include <atomic>
int main(int argc, char **argv)
{
    const int mode = atoi(argv[1]), T = atoi(argv[2]);
    const size_t total = (size_t)atoi(argv[3]) << 30, blk = 24u << 20;
    std::vector<uint8_t> src(blk, 7);
    for (size_t i = 0; i < blk; i += 4096) src[i] = (uint8_t)i;
    unlink("/dev/shm/shmw.bin");
    int fd = open("/dev/shm/shmw.bin", O_RDWR | O_CREAT | O_TRUNC, 0644);
    auto t0 = std::chrono::steady_clock::now();
    if (mode == 1) if (ftruncate(fd, total)) return 1;
    std::atomic<size_t> allocated{0};
    std::thread alloc;
    if (mode == 2)
        alloc = std::thread([&] {
            const size_t step = 256u << 20;
            for (size_t at = 0; at < total; at += step) { if (fallocate(fd, 0, at, std::min(step, total - at))) abort(); allocated = at + std::min(step, total - at); }
        });
    std::vector<std::thread> th;
    const size_t nblk = total / blk;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            for (size_t b = t; b < nblk; b += T) {
                if (mode == 2) while (allocated.load() < (b + 1) * blk) sched_yield();
                if (mode == 0 || mode == 2) { if (pwrite(fd, src.data(), blk, b * blk) != (ssize_t)blk) abort(); }
                else {
                    void *m = mmap(nullptr, blk, PROT_READ | PROT_WRITE, MAP_SHARED, fd, b * blk);
                    if (m == MAP_FAILED) abort();
                    memcpy(m, src.data(), blk);
                    munmap(m, blk);
                }
            }
        });
    for (auto &x : th) x.join();
    if (alloc.joinable()) alloc.join();
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("mode %s threads %d: %.2f GB/s\n", mode == 2 ? "fallocate ahead + pwrite" : mode ? "mmap" : "pwrite", T, total / s / 1e9);
    // read back
    t0 = std::chrono::steady_clock::now();
    th.clear();
    std::vector<std::vector<uint8_t>> dst(T, std::vector<uint8_t>(blk));
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            for (size_t b = t; b < nblk; b += T) {
                if (mode != 1) { if (pread(fd, dst[t].data(), blk, b * blk) != (ssize_t)blk) abort(); }
                else {
                    void *m = mmap(nullptr, blk, PROT_READ, MAP_SHARED, fd, b * blk);
                    memcpy(dst[t].data(), m, blk);
                    munmap(m, blk);
                }
            }
        });
    for (auto &x : th) x.join();
    s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("   read back %s threads %d: %.2f GB/s\n", mode == 1 ? "mmap" : "pread", T, total / s / 1e9);
    close(fd); unlink("/dev/shm/shmw.bin");
}
