// Host-side I/O of the `miekki` binary: gzip-transparent text input and the
// on-disk index format of the reference (Miekki.cpp:649-719),
// streamed through the C ABI in partition ranges.  Plain zlib, no HIP.
#pragma once
#include <string>
#include <vector>

#include "miekki_hip.h"

namespace mkhost {

bool file_exists(const std::string &path);
// whole file; gunzipped when it carries the gzip magic (zstr.hpp:157-167 semantics)
bool read_text(const std::string &path, std::string &out);
// dump_disk: gzip level 1 like zstr::ofstream (zstr.hpp:82), as consecutive 32 MiB
// gzip members compressed by `threads` threads (gzpar.hpp) -- loadable by the
// reference's reader and by zlib's gzread alike.  `ctxs` = the genome shards in id
// order (one context per GPU): their columns side by side are the reference's column.
int dump_index(const std::vector<mk_ctx *> &ctxs, const std::string &path, std::string &err, unsigned threads = 1);
// Miekki(const string&): builds contexts from the file's own header, the genomes split
// in id order over `devices` (never more shards than genomes; every shard receives the
// whole Bloom filter).  Files written by dump_index are inflated by `threads` threads;
// the reference's own dumps (and plain, uncompressed streams) load through gzread.
int load_index(const std::string &path, const std::vector<int> &devices, std::vector<mk_ctx *> &out, std::string &err,
               unsigned threads = 1);

}  // namespace mkhost
