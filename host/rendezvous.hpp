// How the ranks of a one-process-per-GPU run (torch.distributed.run --no-python miekki ..., or MIEKKI_RANK / MIEKKI_WORLD
// by hand) hand rank 0's communicator id to the others: a small file -- no network code in the driver -- that cannot be
// mistaken for another run's and cannot be planted:
//   * its name carries the run's nonce (the launcher's pid and start time, MASTER_PORT, TORCHELASTIC_RUN_ID; or
//     MIEKKI_COMM_NONCE), in $XDG_RUNTIME_DIR, else $TMPDIR, else /tmp -- or it is MIEKKI_COMM_FILE;
//   * rank 0 removes whatever lies under that name, writes the file under a temporary name it creates exclusively
//     (O_CREAT | O_EXCL | O_NOFOLLOW, mode 0600) and renames it into place (a rename replaces a planted link, it does
//     not follow it);
//   * the content is: magic, nonce, the writer's pid and start time, the payload, a checksum.  A reader opens with
//     O_NOFOLLOW, wants a regular file of its own user, the magic, ITS nonce, a checksum that holds -- and a writer that
//     is still alive (pid and start time in /proc): a file a crashed run left behind under a fixed MIEKKI_COMM_FILE has a
//     dead writer, is ignored, and the reader goes on waiting for this run's.  (Ranks that cannot see each other's /proc
//     entries -- a container per rank, hidepid -- would take every writer for dead: MIEKKI_COMM_NO_PID_CHECK=1 leaves that
//     one test out; nonce, owner and checksum still hold);
//   * rank 0 removes the file once every rank has joined the communicator, and at exit whatever happens.
#pragma once
#include <cstddef>
#include <string>

namespace mkhost {

struct Rendezvous {
    std::string path, nonce;
};

// from the environment (see above); `world` only shapes nothing here, the name is the same for every rank of a run
Rendezvous rendezvous_from_env();
bool rendezvous_publish(const Rendezvous &r, const void *payload, size_t n, std::string &err);
// waits up to timeout_ms for this run's file
bool rendezvous_fetch(const Rendezvous &r, void *payload, size_t n, long timeout_ms, std::string &err);
void rendezvous_remove(const Rendezvous &r);

}  // namespace mkhost
