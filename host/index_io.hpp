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
// slice_world > 0: one process per GPU -- this process is rank slice_rank of slice_world, reads the file like every other
// rank and keeps ONE context (on devices[0]) with the columns of its own run of genomes (the split of the -l build:
// contiguous, in id order), the whole Bloom filter, its genomes' sizes.
int load_index(const std::string &path, const std::vector<int> &devices, std::vector<mk_ctx *> &out, std::string &err,
               unsigned threads = 1, int slice_rank = -1, int slice_world = 0);
// -d with one process per GPU: every rank calls it; the ranks' columns travel to rank 0 a block of rows at a time
// (mk_comm_gather) and rank 0 writes the one stream dump_index would have written from all shards in one process.
// Returns the same value on every rank (err: the first failing rank's words).
int dump_index_ranked(mk_ctx *ctx, mk_comm *comm, const std::string &path, std::string &err, unsigned threads = 1);

}  // namespace mkhost
