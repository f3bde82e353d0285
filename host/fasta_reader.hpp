// Ingest pipeline of the host driver: `threads` readers (the reference's -t) read,
// gunzip and parse the listed FASTA files ahead of the consumer, which still receives
// them strictly in list order -- so genome ids and output order stay those of the
// reference at -t 1 (Miekki.cpp:546-581) while the host side keeps up with the device.
//
// A sequence is what Miekki.cpp:559-567 builds: every line that does not start with
// '>' appended to one string.  It is written straight into a buffer from a pool whose
// memory comes from the caller's allocator -- the driver passes mk_host_alloc, i.e.
// pinned memory, so that mk_index_append's copy to the GPU is one DMA per genome with
// no staging copy on the host.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

namespace mkhost {

struct stat_view { bool regular; long long size; };

// CPUs this process may use: the affinity mask, capped by the cgroup's CPU quota (cgroup v2 cpu.max, v1 cfs quota) --
// std::thread::hardware_concurrency() knows neither (a 256-thread host that grants a job 16 CPUs reports 256)
unsigned usable_cpus();

struct HostAllocator {
    void *(*alloc)(void *user, size_t bytes);     // nullptr: malloc / free
    void (*release)(void *user, void *p);
    void *user;
};

// all non-'>' lines of text[0..n) concatenated into dst (room for n bytes); getline semantics
size_t strip_fasta(const char *text, size_t n, char *dst);

// The same sequence, packed as it is parsed (include/miekki_hip.h, mk_packed_seq: 2 bits per base -- base i at
// bits 2 * (i % 32) of word i / 32, A C G T = 0 1 2 3, anything else 0 -- one exception bit per base for anything
// else, the first 32 characters as they came).  The reader has its own packer (32 characters per step, AVX2 when
// the CPU has it, a bit writer that carries over line ends) rather than a call per FASTA line into the library's
// mk_pack_append: at 60-80 characters a line the calls were half of a reader thread's time.
inline size_t packed_code_words(size_t len) { return (len + 31) / 32 + 2; }
inline size_t packed_except_words(size_t len) { return (len + 63) / 64 + 2; }
// returns the sequence's length; *dirty = it holds characters other than A, C, G, T.  text[0..n) must be readable.
size_t pack_fasta(const char *text, size_t n, uint64_t *codes, uint64_t *except, char head[32], bool *dirty);

// the sequence of a gzip'd FASTA file whose bytes are at hand (every member, then strip_fasta): what a raw item becomes
// when the device does not take it
bool inflate_fasta(const char *gz, size_t n, std::vector<char> &seq);

// whole file into `out`, gunzipped when it starts with the gzip magic (any number of members)
bool read_file(const std::string &path, std::vector<char> &out, std::vector<char> &scratch);

// A file's text for one pass over it: a plain file is mapped (no copy out of the page cache), a gzip'd one is
// inflated into `store`.  Closed by the destructor.
struct FileText {
    const char *p = nullptr;
    size_t n = 0;
    bool open(const std::string &path, std::vector<char> &store);
    ~FileText();
private:
    void *map_ = nullptr;
    size_t maplen_ = 0;
};

// Where the gzip'd files of a device unit go when the caller's device takes them piece by piece (the driver binds these to
// mk_gz_open / mk_gz_stage / mk_gz_put): the readers then read every file straight into a page-locked piece the sink lends
// and hand it over -- the files' bytes never wait in host memory, and their way to the device is a DMA per piece that
// nobody waits for.  Without a sink (all null) raw items carry the file's bytes, as before.
struct RawSink {
    void *user = nullptr;
    // a unit of m files of these sizes (0: that file does not go to the device): a batch, or null -- the unit is the readers'
    // then.  offsets[0 .. m]: where each file's bytes lie in the batch's input (file i at offsets[i], room up to offsets[i + 1],
    // what lies between a file's end and the next file is zero): files that follow each other can be put as ONE span
    void *(*open)(void *user, const uint64_t *sizes, uint32_t m, uint64_t *offsets) = nullptr;
    // a buffer of *cap bytes to read into; null: none to be had (the reader uses its own, and says staged = false)
    void *(*stage)(void *user, void *batch, uint64_t *cap) = nullptr;
    // bytes [at, at + bytes) of file i of the batch; staged: `data` came from stage() and is the sink's again
    bool (*put)(void *user, void *batch, uint32_t i, uint64_t at, const void *data, uint64_t bytes, bool staged) = nullptr;
    // files [first, first + count) at once: `data` holds offsets[first + count] - offsets[first] bytes or fewer, laid out as
    // the batch's input is (null: the reader puts file by file)
    bool (*put_span)(void *user, void *batch, uint32_t first, uint32_t count, const void *data, uint64_t bytes, bool staged) = nullptr;
    // every file of the unit has been dealt with (put, or found to be none of the device's): the batch may run -- called by
    // the reader that finished the unit's last file, BEFORE that file's item can be taken, so that the device works on a
    // unit while the consumer is still busy with the units before it
    void (*complete)(void *user, void *batch) = nullptr;
};

class OrderedFastaReader {
public:
    struct Item {
        bool exists = false;
        bool failed = false;       // the file exists but could not be read or buffered: do not treat as empty
        char *data = nullptr;      // pooled buffer: hand it back with recycle()
        size_t len = 0, cap = 0;
        // packed items (a reader made with a PackAppendFn): `data` holds the two arrays, no characters
        bool packed = false, dirty = false;
        // raw items (a reader made with raw_gz): `data` is the FILE as it is -- gzip members, for the device to inflate
        // (mk_gz_unpack) -- len bytes in a pooled buffer like any other item's
        bool raw = false;
        // with a RawSink: every file of a device unit says whose it is -- unit_batch (what the sink's open returned), its place
        // unit_index in it, unit_last on the unit's last file: when that one has been taken, every file of the unit has been
        // put.  A raw item then has no `data` (its bytes are on the device); a file of the unit that did not go there (not
        // gzip after all, unreadable, missing) comes as the ordinary item it would have been.
        void *unit_batch = nullptr;
        uint32_t unit_index = 0;
        bool unit_last = false;
        uint64_t *codes = nullptr, *except = nullptr;
        char head[32] = {0};
    };
    // window = how many files may be parsed ahead of the consumer (each holds one buffer);
    // packed: items come packed (a quarter of the bytes to buffer and to copy to the GPU)
    // raw_gz: files that start with the gzip magic are handed over as they are (Item::raw), everything else as before
    // raw_unit / raw_units_ahead: the gzip'd files are SHARED between the device's inflater and the readers' own zlib --
    // the list is cut into units of raw_unit files; a unit goes to the device (its files come raw) while fewer than
    // raw_units_ahead units' worth of raw files are waiting for the consumer's raw_consumed(), and is inflated here
    // otherwise.  raw_unit 0: every gzip'd file raw.
    // sink + share: with a sink the list is always cut into units of raw_unit files; share = false: every unit whose first
    // file is gzip'd goes to the device (the read-ahead window is what bounds the units in flight)
    OrderedFastaReader(std::vector<std::string> files, unsigned threads, HostAllocator a, size_t window, bool packed = false,
                       bool raw_gz = false, size_t raw_unit = 0, size_t raw_units_ahead = 0, RawSink sink = RawSink(), bool share = true);
    // the consumer is done with n raw items (appended or given up)
    void raw_consumed(size_t n) { raw_out_.fetch_sub((long)n); }
    ~OrderedFastaReader();
    // blocks until file i (called with i = 0, 1, 2, ...) has been read
    Item take(size_t i);
    void recycle(Item &it);
private:
    void work();
    char *pool_get(size_t need, size_t &cap, bool plain = false);   // plain: ordinary memory even when there is an allocator
    void pool_release(char *p);                    // caller holds pool_m_ or is the destructor
    std::vector<std::string> files_;
    std::vector<Item> items_;
    std::vector<std::atomic<int>> ready_;
    std::vector<std::thread> workers_;
    std::atomic<size_t> next_{0};
    size_t consumed_ = 0, window_ = 8, ahead_bytes_ = 0;
    bool stop_ = false;                            // set by the destructor: parked workers leave
    size_t ahead_limit_ = 4ull << 30;
    std::mutex m_;
    std::condition_variable cv_;
    HostAllocator a_;
    bool pack_ = false, raw_gz_ = false;
    size_t raw_unit_ = 0;
    long raw_limit_ = 0;
    std::vector<std::atomic<int>> umode_;          // per unit: 0 undecided, 1 to the device, 2 inflated here, 3 being opened
    std::vector<void *> ubatch_;                   // per unit: the sink's batch (mode 1 with a sink)
    std::vector<std::atomic<uint32_t>> udone_;     // per unit: files dealt with
    std::vector<std::vector<uint64_t>> usize_, uoff_;   // per unit (mode 1 with a sink): the files' sizes as the batch knows them, their places
    size_t run_ = 1;                               // files a reader takes at a time (a run of a device unit goes up as one span)
    struct RunStage;                               // a reader's span being filled
    bool stage_raw(size_t i, size_t unit, const stat_view &sv, Item &it, RunStage &rs);
    void flush_stage(RunStage &rs);
    void publish(size_t i, const Item &it);
    void unit_file_done(size_t unit, void *batch, uint32_t k);
    RawSink sink_;
    bool share_ = true;
    bool put_raw(size_t i, const struct stat_view &sv, Item &it);
    std::atomic<long> raw_out_{0};                 // files of device units handed out or still to come, not yet consumed
    std::mutex pool_m_;
    std::vector<std::pair<char *, size_t>> pool_;   // free buffers (pointer, capacity)
    std::unordered_set<char *> plain_;              // buffers that came from malloc although an allocator was given
};

}  // namespace mkhost
