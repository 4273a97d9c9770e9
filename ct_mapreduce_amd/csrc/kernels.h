// kernels.h — the hand-written HIP kernels of the library, gfx950 (CDNA4, wave64) only.  No MFMA anywhere: this path
// is byte/integer scan + hash work bounded by HBM bandwidth.
//
//   kernels/readers.h   GlobalReader, WinReader/C/S — per-lane 256-byte LDS windows, wave-cooperative fills
//   kernels/sha256.h    k_issuer_ids (issuer table: walk Chain[0], SHA-256(RawSubjectPublicKeyInfo)), k_sha256_one
//   kernels/meta_core.h the read-only side of the IssuerMetadata memo (slot layout, item hash, CRL-DP walk, the map's pre-check)
//   kernels/map.h       the map — map_one (walk + filters + record), k_map_winc (the map without the fused insert)
//   kernels/map_sweep.h k_map_tile / k_map_direct: the two baseline designs, CTMR_SWEEP builds only (libctmr_sweep.so)
//   kernels/reduce.h    the known-certificate table (8-byte index words + 64-byte key cells): index_upsert, k_insert /
//                       k_insert2 (settle_order), k_map_fused (THE dominant kernel: walk + key parse + filters + pass 1 of
//                       the insert from the walking lane's registers; the default), k_ec_resolve (the curve equation of EC
//                       keys, then their pass 1), k_resolve, k_scan_blocks, k_compact
//   kernels/exchange.h  k_key_blockcount / k_key_gather / k_keys_insert* / k_keys_resolve / k_apply_lost (owner-computes
//                       exchange), k_filter_interleave / k_bloom_probe / k_keys_lookup / k_bloom_apply (Bloom variant)
//   kernels/pem.h       k_pem_len, k_pem_encode
//   kernels/entries.h   k_decode_match (decode + first Chain[0] match round), k_chain0_match, k_entry_decode (sweep builds)
//   kernels/meta.h      k_meta_new
//   kernels/misc.h      k_fingerprint, k_set_op / k_sweep / k_rehash / k_arena_compact / k_build_pairs / k_list / k_pairs,
//                       k_synth_*
// der_walk.h is the TBSCertificate walk every kernel above shares; spki_key.h the key inside SubjectPublicKeyInfo.
#pragma once
#include "kernels/readers.h"
#include "kernels/sha256.h"
#include "kernels/meta_core.h"
#include "kernels/map.h"
#ifdef CTMR_SWEEP
#include "kernels/map_sweep.h"
#endif
#include "kernels/reduce.h"
#include "kernels/exchange.h"
#include "kernels/pem.h"
#include "kernels/entries.h"
#include "kernels/meta.h"
#include "kernels/misc.h"
