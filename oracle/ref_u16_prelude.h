/* Forced-include prelude for the 16-bit reference build.
 *
 * The reference selects its fingerprint storage type with one compile-time
 * token, `#define minimizer uint8_t` (utils.h:27).  Its 2-byte code path
 * (Miekki.cpp:229-231, 883-890) is only coherent when that token is uint16_t.
 * This prelude includes the reference's own utils.h first (so its include guard
 * is set and the later includes are no-ops) and then re-points the token; the
 * reference sources themselves are compiled unmodified from /root/reference.
 */
#include "utils.h"
#undef minimizer
#define minimizer uint16_t
