/* umgap_hip.h — C ABI of libumgap_hip.so: the MI355X (gfx950) implementation of UMGAP's
 * pept2lca -> taxa2agg hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * The reference (unipept/umgap v1.1.1, Rust) has no FFI for this path: the path sits behind
 * two in-process seams, which these entry points replace one for one (file:line relative to
 * the reference tree):
 *
 *   fst::Map::{from_path,from_bytes}   src/commands/pept2lca.rs:74-79     -> umgap_index_open / _build9
 *   fst::Map::get(&[u8])->Option<u64>  src/commands/pept2lca.rs:93,
 *                                      src/commands/prot2kmer2lca.rs:176  -> umgap_index_lookup9[_dev]
 *   fst::MapBuilder::{insert,finish}   src/commands/buildindex.rs:38-45   -> umgap_index_build9 + _save
 *   taxon::{TaxonList,TaxonTree}::new,
 *   TaxonTree::snapping                src/commands/taxa2agg.rs:104-109   -> umgap_taxonomy_create
 *   trait agg::Aggregator::aggregate   src/agg/mod.rs:12-21 (+ count/filter :27-44 and the
 *                                      record loop src/commands/taxa2agg.rs:161-180)
 *                                                                         -> umgap_aggregate[_dev]
 *   seedextend record state machine    src/commands/seedextend.rs:99-148  -> umgap_seedextend_dev
 *   the four 9-mer stages of scripts/umgap-analyse.sh:257-310 in one call (hits never leave the device)
 *                                                                         -> umgap_classify9[_strategy]_dev,
 *                                                                            umgap_classify_prot9[_strategy]_dev,
 *                                                                            umgap_classify_records (host records)
 *
 * Conventions
 *   - every function returns UMGAP_OK (0) or a negative UMGAP_ERR_* code; the message is in the
 *     thread-local umgap_last_error().  Nothing throws or aborts across this boundary.
 *   - the caller owns every buffer it passes; the library owns device memory and streams.
 *   - `_dev` variants take DEVICE pointers (resident in the handle's GPU's HBM) and a hipStream_t
 *     passed as void* (NULL = default stream); they enqueue and return without synchronising.
 *     The plain variants take HOST pointers and return when the result is in the caller's memory:
 *     big calls run in chunks, uploads ahead of the kernels; arrays in pinned memory (umgap_host_alloc)
 *     are copied from / to directly, pageable ones through pinned staging buffers.
 *   - handles are thread-compatible: one in-flight call per handle; distinct handles may be used
 *     from distinct threads.
 *   - there is NO CPU fallback: without a usable GPU every compute entry point fails with
 *     UMGAP_ERR_DEVICE.
 *
 * 9-mer packing convention: key45 = sum_i code(c_i) << (5*(8-i)), code = c-'A'
 * for 'A'..'Z'.  A peptide holding any other byte is not representable: the host emits the miss
 * value without calling the device (umgap_pack9 returns 0 for it).
 */
#ifndef UMGAP_HIP_H
#define UMGAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UMGAP_OK 0
#define UMGAP_ERR_INVALID (-1)    /* bad argument */
#define UMGAP_ERR_IO (-2)         /* file could not be read/written */
#define UMGAP_ERR_FORMAT (-3)     /* malformed input (index file, taxonomy) */
#define UMGAP_ERR_DEVICE (-4)     /* no GPU / HIP failure */
#define UMGAP_ERR_NOMEM (-5)
#define UMGAP_ERR_BUILD (-6)      /* index build failed (duplicate key, table too full) */
#define UMGAP_ERR_PANIC (-7)      /* the reference would panic here (exit 101), e.g. "More than one root!" */

/* per-read status written by umgap_aggregate (src/commands/taxa2agg.rs:161-180 outcomes) */
#define UMGAP_READ_OK 0            /* out_taxon = snapping[aggregate]                       */
#define UMGAP_READ_EMPTY 1         /* no taxa left after the filters: out_taxon = 1 (:174)  */
#define UMGAP_READ_UNKNOWN_TAXON 2 /* Err(UnknownTaxon(out_taxon)): host aborts, exit 1     */
#define UMGAP_READ_PANIC 3         /* index out of bounds / unwrap on None: host exits 101  */
#define UMGAP_READ_UNSUPPORTED 4   /* > 32 768 distinct taxa in one read (hybrid/mrtl only): not computed */

#define UMGAP_METHOD_TREE 0 /* -m tree (src/commands/taxa2agg.rs:186-190) */
#define UMGAP_METHOD_RMQ 1  /* -m rmq  */
#define UMGAP_STRATEGY_LCA_STAR 0 /* -a lca*   */
#define UMGAP_STRATEGY_HYBRID 1   /* -a hybrid */
#define UMGAP_STRATEGY_MRTL 2     /* -a mrtl   */

#define UMGAP_NONE 0xFFFFFFFFu

typedef struct umgap_index umgap_index;
typedef struct umgap_taxonomy umgap_taxonomy;

/* ---- library --------------------------------------------------------------------------- */
const char* umgap_last_error(void);
const char* umgap_version(void);      /* "1.1.1" — the reference version mirrored */
int umgap_device_count(void);         /* number of visible GPUs, 0 if none (never fails) */

/* Packs one 9-mer; returns 1 and writes *key45, or 0 if any byte is outside 'A'..'Z'. */
int umgap_pack9(const char* nine_bytes, uint64_t* key45);
void umgap_unpack9(uint64_t key45, char* nine_bytes);

/* ---- index: 9-mer -> taxon table resident in HBM --------------------------------------- */
/* Builds a table on `device` from host arrays (keys distinct, any order).  load_factor in
 * (0,1): filled fraction of the 8-slot buckets (0 selects the default 0.5 = 16 B per key: 2.3 % of
 * the buckets overflow; 6.6 % at 0.6). */
int umgap_index_build9(const uint64_t* keys45, const uint32_t* taxa, size_t n, double load_factor,
                       int device, umgap_index** out);
/* Same, key/value arrays already in device memory (on `device`).  Every build that takes device arrays (this one,
 * umgap_index_build_var_dev, umgap_index_build_windows9_dev) first waits for the device: the arrays may have been produced on
 * any stream. */
int umgap_index_build9_dev(const uint64_t* d_keys45, const uint32_t* d_taxa, size_t n, double load_factor,
                           int device, umgap_index** out);
/* Writes / reads the table image ("UMGAPIX1" format).  device_mask: bit d set =
 * GPU d may be used; the lowest set bit is chosen (one GPU per handle, one process per GPU).
 * shard/n_shards: load only that contiguous bucket range (+ overflow halo); 0/1 = whole table. */
int umgap_index_save(const umgap_index* ix, const char* path);
int umgap_index_open(const char* path, uint32_t device_mask, umgap_index** out);
int umgap_index_open_shard(const char* path, uint32_t device_mask, uint32_t shard, uint32_t n_shards,
                           umgap_index** out);
/* Restrict an already built whole-table handle to one shard (frees nothing; lookups of keys whose
 * home bucket lies outside the shard return `miss_value`).  Used by the sharded multi-GPU mode. */
int umgap_index_set_shard(umgap_index* ix, uint32_t shard, uint32_t n_shards);
/* Owner shard of each key under an n_shards split (for the all-to-all exchange). */
int umgap_index_owner9_dev(const umgap_index* ix, const uint64_t* d_keys45, size_t n, uint32_t n_shards,
                           uint32_t* d_owner, void* stream);
/* Sharded mode, send side: scatters the keys into `d_send` grouped by owner shard (n_shards
 * regions of `cap` keys; the caller pre-fills d_send with an invalid key such as ~0 and zeroes
 * d_cursor[n_shards] and *d_overflow) and records in d_perm[i] where key i went, so that the
 * returned taxa can be gathered back.  *d_overflow != 0 afterwards = some region exceeded cap. */
int umgap_index_partition9_dev(const umgap_index* ix, const uint64_t* d_keys45, size_t n, uint32_t n_shards,
                               uint64_t cap, uint64_t* d_send, int32_t* d_perm, uint64_t* d_cursor,
                               uint32_t* d_overflow, void* stream);
/* Incremental build from host chunks, for inputs that do not fit host memory (`umgap buildindex` streams a TSV
 * of any size through it; src/commands/buildindex.rs:32-48): the table is sized for expected_keys at the given
 * load, chunks are uploaded and inserted as they come, finish() reports duplicates / overflow and hands over the
 * index (the builder is consumed).  The caller guarantees nothing about key order. */
typedef struct umgap_index_builder umgap_index_builder;
int umgap_index_builder_new(uint64_t expected_keys, double load_factor, int device, umgap_index_builder** out);
int umgap_index_builder_add(umgap_index_builder* b, const uint64_t* keys45, const uint32_t* taxa, size_t n);
int umgap_index_builder_finish(umgap_index_builder* b, umgap_index** out);
void umgap_index_builder_free(umgap_index_builder* b);
void umgap_index_free(umgap_index* ix);
uint64_t umgap_index_num_keys(const umgap_index* ix);
uint64_t umgap_index_num_buckets(const umgap_index* ix);
uint64_t umgap_index_table_bytes(const umgap_index* ix);
int umgap_index_device(const umgap_index* ix);

/* get(): out_taxa[i] = value of keys45[i], or miss_value.  keys with bits above 45 set miss. */
int umgap_index_lookup9(umgap_index* ix, const uint64_t* keys45, size_t n, uint32_t miss_value,
                        uint32_t* out_taxa);
int umgap_index_lookup9_dev(umgap_index* ix, const uint64_t* d_keys45, size_t n, uint32_t miss_value,
                            uint32_t* d_out_taxa, void* stream);
/* Fused prot2kmer2lca surface (src/commands/prot2kmer2lca.rs:172-179): `bytes` holds n_prot
 * peptides, peptide p = bytes[offsets[p] .. offsets[p+1]); every window of k=9 residues is
 * looked up; window w of peptide p lands in out[out_offsets[p] + w] where out_offsets[p] =
 * sum over q<p of max(0, len_q - 8) (the caller computes it; device pointers). Windows that
 * contain a byte outside 'A'..'Z' get miss_value. */
int umgap_index_lookup_prot9_dev(umgap_index* ix, const uint8_t* d_bytes, const uint64_t* d_offsets,
                                 const uint64_t* d_out_offsets, size_t n_prot, uint32_t miss_value,
                                 uint32_t* d_out_taxa, void* stream);

/* ---- variable-length keys (tryptic peptides, 1..63 bytes) -------------------------------- */
/* The same handle type; umgap_index_kind() tells 9 (packed 9-mer table) from 2 (byte-string table) and 10 (9-mer table in walk order).
 * Key i = bytes[offsets[i] .. offsets[i+1]).  Exact match.
 * Replaces fst::Map::get for the tryptic index (src/commands/pept2lca.rs:93,
 * src/commands/prot2tryp2lca.rs:130).  umgap_index_save / _open / _free work for both kinds.
 * A byte-string table has one of two layouts, chosen at build time: when every key is a word over 'A'..'Z' (every tryptic
 * index is) the keys themselves are stored in the table at 5 bits per residue, 1-3 adjacent 16-byte slots each, and a lookup is
 * one 64-byte request, hit or miss (umgap_index_var_layout() = 1); otherwise slots point into a heap of key bytes and a
 * candidate is verified byte by byte there (layout 0; the environment variable UMGAP_VAR_PACKED=0 forces it). */
int umgap_index_kind(const umgap_index* ix);
int umgap_index_var_layout(const umgap_index* ix);   /* 1 = keys in the slots, 0 = slots + heap, -1 = not a byte-string table */
int umgap_index_build_var(const uint8_t* bytes, const uint64_t* offsets, const uint32_t* taxa, size_t n,
                          double load_factor, int device, umgap_index** out);
int umgap_index_build_var_dev(const uint8_t* d_bytes, const uint64_t* d_offsets, const uint32_t* d_taxa, size_t n,
                              uint64_t n_bytes, double load_factor, int device, umgap_index** out);
int umgap_index_lookup_var(umgap_index* ix, const uint8_t* bytes, const uint64_t* offsets, size_t n,
                           uint32_t miss_value, uint32_t* out_taxa);
int umgap_index_lookup_var_dev(umgap_index* ix, const uint8_t* d_bytes, const uint64_t* d_offsets, size_t n,
                               uint32_t miss_value, uint32_t* d_out_taxa, void* stream);

/* ---- tryptic digest on the device: the front half of `prot2tryp2lca` (src/commands/prot2tryp2lca.rs:107-131) ----------- */
/* The default cleavage pattern `([KR])([^P])` as a per-byte rule: a piece starts where a fragment starts, after a K or R that
 * is not followed by P, and at and after every `*`.  Fragment f = d_bytes[d_frag_offsets[f] .. d_frag_offsets[f+1]) (offsets
 * must not decrease).  Every byte belongs to exactly one piece and pieces lie back to back, so the digest is the sorted list of
 * piece starts plus the end: piece i = d_bytes[d_piece_offsets[i] .. d_piece_offsets[i+1]) -- the (bytes, offsets) key list
 * umgap_index_lookup_var_dev takes as it is.  A `*` is a piece of its own (one byte); empty lines do not arise.
 * *n_pieces is always set; the offsets (n_pieces + 1 of them) are written only if cap >= n_pieces + 1: call once with cap = 0 to
 * size the buffer.  Synchronous (the count makes a host round trip).  A custom -p pattern stays with the host (umgap CLI). */
int umgap_digest_tryp_dev(const uint8_t* d_bytes, const uint64_t* d_frag_offsets, size_t n_fragments,
                          uint64_t* d_piece_offsets, size_t cap, size_t* n_pieces, int device, void* stream);
/* The tryptic pipeline of scripts/umgap-analyse.sh (prot2tryp2lca -l min -L max | taxa2agg ...) for device-resident reads:
 * read r owns fragments [r*F, (r+1)*F); digest as above, pieces of min_len..max_len bytes are looked up in the variable-length
 * index (every other piece, and every `*`, is no hit), and the hits of a read are aggregated as umgap_aggregate_dev does.
 * -k / -d (contains / lacks) are not offered here.  The digest's count makes a host round trip; everything after it is
 * stream-ordered and the call returns without waiting for it.  Scratch lives in the index handle (grow-only): issue the calls
 * of one handle on one stream.  *n_pieces_out (nullable) = pieces the digest produced. */
int umgap_classify_tryp_dev(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound,
                            int ranked_only, const uint8_t* d_bytes, const uint64_t* d_frag_offsets, size_t n_reads,
                            uint32_t fragments_per_read, uint32_t min_len, uint32_t max_len, uint32_t* d_out_taxon,
                            uint32_t* d_out_status, uint64_t* n_pieces_out, void* stream);

/* ---- dump: fst::Map::stream() as used by `umgap printindex` (src/commands/printindex.rs:43-47) ---- */
/* Every (key, taxon) of a whole table, in table order (the caller sorts).  *n_out = number of keys
 * (also when it exceeds cap; then only the first cap are written). */
int umgap_index_dump9(const umgap_index* ix, uint64_t* keys45, uint32_t* taxa, size_t cap, size_t* n_out);
/* Variable-length table: key i = heap_out[offsets_out[i] .. + lens_out[i]).  Call once with heap_out = NULL to learn
 * *heap_bytes_out (the key heap plus the keys the table holds inside its slots, which are unpacked behind it), then again
 * with a buffer of that size. */
int umgap_index_dump_var(const umgap_index* ix, uint8_t* heap_out, uint64_t heap_cap, uint64_t* offsets_out,
                         uint32_t* lens_out, uint32_t* taxa_out, size_t cap, size_t* n_out, uint64_t* heap_bytes_out);

/* ---- taxonomy ---------------------------------------------------------------------------- */
/* rows in TSV order: id, parent, valid (0/1), rank (index into src/rank.rs:10-41, 0 = "no rank").
 * Children keep row order (Euler tour / preorder follow it, src/taxon.rs:224-247).
 * Errors: UMGAP_ERR_PANIC for the reference's panics (no rows, "More than one root!",
 * "There's no root!"); UMGAP_ERR_FORMAT for ids >= 2^32-1, the same id under two different parents, or parent cycles
 * (the reference loops forever on a cycle).  An id repeated with the same parent is accepted as the reference takes it: the
 * last row's rank / valid, the first row's place among its siblings. */
int umgap_taxonomy_create(const uint32_t* id, const uint32_t* parent, const uint8_t* valid,
                          const uint8_t* rank, size_t n, int device, umgap_taxonomy** out);
void umgap_taxonomy_free(umgap_taxonomy* tx);
uint32_t umgap_taxonomy_root(const umgap_taxonomy* tx);
uint32_t umgap_taxonomy_len(const umgap_taxonomy* tx);     /* max id + 1 (TaxonList length)    */
uint32_t umgap_taxonomy_reachable(const umgap_taxonomy* tx); /* nodes reachable from the root  */
/* host copies of the derived per-id tables (len entries each; UMGAP_NONE = None) */
int umgap_taxonomy_snapping(const umgap_taxonomy* tx, int ranked_only, uint32_t* out);
int umgap_taxonomy_preorder(const umgap_taxonomy* tx, uint32_t* pre, uint32_t* last, uint32_t* depth);

/* ---- aggregate: one consensus taxon per read ------------------------------------------------ */
/* taxa/scores are CSR over reads: read r owns taxa[read_offsets[r] .. read_offsets[r+1]).
 * Per read (src/commands/taxa2agg.rs:161-180): drop taxon 0, sum scores per taxon (1.0 each
 * when scores == NULL), keep sums >= lower_bound, aggregate with (method, strategy, factor),
 * snap to the nearest valid (and ranked if ranked_only) ancestor.
 * Ties that the reference breaks by HashMap order are broken by smallest preorder index. */
int umgap_aggregate(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound,
                    int ranked_only, const uint32_t* taxa, const float* scores,
                    const uint64_t* read_offsets, size_t n_reads, uint32_t* out_taxon,
                    uint32_t* out_status);
int umgap_aggregate_dev(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound,
                        int ranked_only, const uint32_t* d_taxa, const float* d_scores,
                        const uint64_t* d_read_offsets, size_t n_reads, uint32_t* d_out_taxon,
                        uint32_t* d_out_status, void* stream);
/* Fixed-stride form (every read owns `stride` consecutive taxa, 0 = padding/miss): the layout
 * the fused 9-mer pipeline produces; avoids the offsets array. */
int umgap_aggregate_strided_dev(umgap_taxonomy* tx, int method, int strategy, float factor,
                                float lower_bound, int ranked_only, const uint32_t* d_taxa,
                                uint32_t stride, size_t n_reads, uint32_t* d_out_taxon,
                                uint32_t* d_out_status, void* stream);

/* ---- seedextend on device (src/commands/seedextend.rs:99-148; ranked mode :150-163 below) ---- */
/* In place per record r = taxa[offsets[r]..offsets[r+1]): taxa outside every extended seed
 * are overwritten with 0 (a dropped taxon and a 0 are equivalent for taxa2agg, which discards
 * zeros, src/commands/taxa2agg.rs:169).  kept[r] (nullable) receives the number of taxa the
 * reference's seedextend would print for the record, or UMGAP_NONE where the reference panics
 * (exit 101): a seed sliced with start > end (src/commands/seedextend.rs:169-173), which needs
 * min_seed_size < 2 and a leading run of misses; such a record is zeroed entirely.  The fused
 * kernels below report the same case as UMGAP_READ_PANIC for the read. */
int umgap_seedextend_dev(const uint64_t* d_offsets, size_t n_records, uint32_t min_seed_size,
                         uint32_t max_gap_size, uint32_t* d_taxa, uint32_t* d_kept, int device,
                         void* stream);
/* Ranked mode, `seedextend -r taxa.tsv -p penalty` (src/commands/seedextend.rs:77-83,150-163): of a record's extended seeds only
 * the one with the highest score stays -- the last of equal ones, as Iterator::max_by_key -- a seed's score being the sum over
 * its taxa of TaxonList::score (12 below a named rank, else None) with `penalty` (the command's default: 5) for None and for ids
 * the taxonomy's list does not hold.  The scores come from the rows `tx` was made of (rank column); everything else as
 * umgap_seedextend_dev.  The fused pipelines run the unranked mode (no script uses -r). */
int umgap_seedextend_ranked_dev(umgap_taxonomy* tx, const uint64_t* d_offsets, size_t n_records, uint32_t min_seed_size,
                                uint32_t max_gap_size, uint32_t penalty, uint32_t* d_taxa, uint32_t* d_kept, void* stream);
/* Fixed-stride form: record r = d_taxa[r*stride .. (r+1)*stride). */
int umgap_seedextend_strided_dev(uint32_t stride, size_t n_records, uint32_t min_seed_size,
                                 uint32_t max_gap_size, uint32_t* d_taxa, uint32_t* d_kept, int device,
                                 void* stream);

/* ---- fused 9-mer pipeline ------------------------------------------------------------------ */
/* One kernel for `prot2kmer2lca -o | seedextend -g G -s S | uniq -d / | taxa2agg -l L -a lca*`
 * (scripts/umgap-analyse.sh:299-310) over reads given as packed 9-mers: read r owns
 * d_keys45[r*kmers_per_read .. +kmers_per_read) (<= 128).  fragments_per_read = 0 skips
 * seedextend; otherwise each read is that many equal fragments (2 = the two ends of a pair) and
 * seedextend (unranked) runs per fragment.  LCA* only; hits never leave the chip. */
int umgap_classify9_dev(umgap_index* ix, umgap_taxonomy* tx, int method, float lower_bound, int ranked_only,
                        const uint64_t* d_keys45, size_t n_reads, uint32_t kmers_per_read,
                        uint32_t fragments_per_read, uint32_t min_seed_size, uint32_t max_gap_size,
                        uint32_t* d_out_taxon, uint32_t* d_out_status, void* stream);

/* Binding the table to a taxonomy: the slot values are rewritten in place (about 0.3 s for 128 GB) from taxon
 * ids to {preorder index, subtree size code}, so that umgap_classify9_dev with the SAME taxonomy handle aggregates
 * without gathering the by-id table (fewer random requests; results unchanged).  Refused, with the table left as
 * it was, when some value is not a reachable taxon of that taxonomy (the unbound kernels report those per read)
 * or the taxonomy has more than 2^22 - 2 reachable taxa.  The fused calls -- umgap_classify9[_strategy]_dev,
 * umgap_classify_prot9[_strategy]_dev and umgap_classify_records (what `umgap prot2kmer2agg` runs: it binds once at
 * start-up) -- take a handle bound to the taxonomy they are given; while bound, every other lookup / dump / save
 * entry point refuses the handle; umgap_index_unbind_taxonomy restores the taxon ids.
 * Also the peptide table in its packed layout (umgap_index_var_layout() == 1: every key a word over 'A'..'Z'; round 4): the taxon
 * word of every entry becomes the code, umgap_classify_tryp_dev then aggregates without a by-id gather (reads its first pass
 * defers get taxon ids back for the large-scratch passes), the plain look-ups and the dump refuse the handle until it is unbound. */
int umgap_index_bind_taxonomy(umgap_index* ix, umgap_taxonomy* tx);
int umgap_index_unbind_taxonomy(umgap_index* ix, umgap_taxonomy* tx);
int umgap_index_is_bound(const umgap_index* ix);

/* The same for every aggregation strategy of taxa2agg (MRTL, tree-hybrid = its default, rmq-hybrid;
 * UMGAP_STRATEGY_LCA_STAR forwards to umgap_classify9_dev): the fused form of
 * `prot2kmer2lca -o | seedextend | uniq -d / | taxa2agg -m M -a A -f F -l L` for the sensitivity
 * pipelines (scripts/umgap-analyse.sh:277-288).  Ties as in umgap_aggregate (smallest preorder). */
int umgap_classify9_strategy_dev(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor,
                                 float lower_bound, int ranked_only, const uint64_t* d_keys45, size_t n_reads,
                                 uint32_t kmers_per_read, uint32_t fragments_per_read, uint32_t min_seed_size,
                                 uint32_t max_gap_size, uint32_t* d_out_taxon, uint32_t* d_out_status, void* stream);

/* Surface B of the same pipeline: the reads arrive as protein fragments (1 byte per residue, what
 * `prot2kmer2lca` reads, src/commands/prot2kmer2lca.rs:172-179): read r owns fragments
 * [r*F, (r+1)*F) of the CSR (d_bytes, d_frag_offsets[n_reads*F + 1]); every window of 9 residues of
 * every fragment is looked up (fragments shorter than 9 have none; windows with a byte outside
 * 'A'..'Z' miss).  Reads of 1..4 fragments: a read with more than 128 windows or 384 bytes gets status
 * UMGAP_READ_UNSUPPORTED (use umgap_index_lookup_prot9_dev + umgap_aggregate_dev for those).  Reads of 5..32 fragments -- a pair
 * translated in all six frames (`translate -a`, scripts/umgap-analyse.sh:277-288: 12 fragments, ~500 windows) -- take the LONG
 * variant of the kernel: up to 512 windows and 1536 bytes per read, of which at most 128 may still carry a hit when the
 * aggregate stage is reached (after seedextend; the hits of such a read sit in the one or two frames that code), else
 * UMGAP_READ_UNSUPPORTED. */
int umgap_classify_prot9_dev(umgap_index* ix, umgap_taxonomy* tx, int method, float lower_bound, int ranked_only,
                             const uint8_t* d_bytes, const uint64_t* d_frag_offsets, size_t n_reads,
                             uint32_t fragments_per_read, int seedextend, uint32_t min_seed_size,
                             uint32_t max_gap_size, uint32_t* d_out_taxon, uint32_t* d_out_status, void* stream);
/* Surface B with any aggregation strategy (UMGAP_STRATEGY_LCA_STAR forwards to the call above; `factor` as in
 * umgap_aggregate).  For MRTL and the hybrids a read with more than 64 distinct taxa surviving the count filter goes through a
 * large-scratch pass behind the first launch (round 4); more than 128 is UMGAP_READ_UNSUPPORTED. */
int umgap_classify_prot9_strategy_dev(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor,
                                      float lower_bound, int ranked_only, const uint8_t* d_bytes,
                                      const uint64_t* d_frag_offsets, size_t n_reads, uint32_t fragments_per_read,
                                      int seedextend, uint32_t min_seed_size, uint32_t max_gap_size,
                                      uint32_t* d_out_taxon, uint32_t* d_out_status, void* stream);

/* The same pipeline for a block of FASTA records held in HOST memory -- what a fused
 * `prot2kmer2lca -o | seedextend -s S -g G | uniq [-d D] | taxa2agg ...` command needs per block
 * (scripts/umgap-analyse.sh:257-310): fragment f = bytes[frag_offsets[f] .. frag_offsets[f+1]) is the
 * sequence of one record prot2kmer2lca would print (src/commands/prot2kmer2lca.rs:172-179; the caller
 * drops records shorter than 9 like :172 does), read r = fragments [read_frags[r], read_frags[r+1]) are
 * the consecutive records `uniq` merges (src/commands/uniq.rs:66-79).  Every 9-residue window is
 * looked up (miss = 0 as with -o), seedextend (unranked, src/commands/seedextend.rs:99-148) runs per
 * fragment when `seedextend` != 0, and the read's surviving hits are aggregated exactly as
 * umgap_aggregate does (any method / strategy; out_status as there).  Reads of any length. */
/* out_status additionally takes UMGAP_READ_PANIC for a read holding a fragment on which the reference's seedextend would
 * panic (a seed sliced with start > end, src/commands/seedextend.rs:169-173; only with min_seed_size < 2).  The block is
 * processed in chunks of whole reads, chunks c+1 and c+2 uploading while chunk c computes; pass arrays from umgap_host_alloc:
 * they are copied from / to directly (383 M pairs/s of 2 x 50 residues with the link included); pageable arrays go through
 * pinned staging with a host memcpy (302 M). */
int umgap_classify_records(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor,
                           float lower_bound, int ranked_only, const uint8_t* bytes,
                           const uint64_t* frag_offsets, size_t n_fragments, const uint64_t* read_frags,
                           size_t n_reads, int seedextend, uint32_t min_seed_size, uint32_t max_gap_size,
                           uint32_t* out_taxon, uint32_t* out_status);
/* The same call with the residues in PIECES laid end to end (round 5): piece i holds the bytes [piece_ends[i - 1], piece_ends[i]) of
 * the array `bytes` would have been (piece_ends[-1] = 0; frag_offsets index that concatenation; a fragment may run across pieces).
 * A host that tokenizes a block of FASTA on several threads has one residue buffer per thread: pieces from umgap_host_alloc are
 * copied to the device from where they lie (`umgap prot2kmer2agg` saves the memcpy that gathered them: 16 M pairs stream through
 * in 0.26-0.29 s instead of 0.34-0.37 s, profiles/r05_fused_pieces_ab.json); pageable pieces are gathered into the pinned staging,
 * one threaded memcpy as before. */
int umgap_classify_records_sg(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor,
                              float lower_bound, int ranked_only, const uint8_t* const* pieces,
                              const uint64_t* piece_ends, size_t n_pieces, const uint64_t* frag_offsets,
                              size_t n_fragments, const uint64_t* read_frags, size_t n_reads, int seedextend,
                              uint32_t min_seed_size, uint32_t max_gap_size, uint32_t* out_taxon,
                              uint32_t* out_status);
/* What `prot2kmer2lca` does per record (src/commands/prot2kmer2lca.rs:172-179: every 9-residue window of the record through
 * fst::Map::get) for a whole block of records in ONE call, the windows cut on the device: fragment f = bytes[frag_offsets[f] ..
 * frag_offsets[f+1]) (host memory; pinned arrays are copied from / to directly), out_hits[b - frag_offsets[0]] = value of the 9-mer
 * that starts at byte b for every b with b + 9 <= end of its fragment, miss_value where that window is not [A-Z]{9} or not a key,
 * and at the positions that start no window (the last 8 of a fragment).  1.2 bytes per window cross the link instead of an 8-byte
 * packed key, and the host does no per-window work.  `umgap prot2kmer2lca` on a 9-mer index runs on this call. */
int umgap_index_lookup_records(umgap_index* ix, const uint8_t* bytes, const uint64_t* frag_offsets, size_t n_fragments,
                               uint32_t miss_value, uint32_t* out_hits);
/* Which kernels the last umgap_classify_records of the calling thread ran: 1 = one fused launch per chunk (every read of the call
 * has the same number of fragments, at most 32, and fits the limits of umgap_classify_prot9_strategy_dev -- its short or its LONG
 * variant, picked per chunk on the device: what paired reads look like, as gene fragments or translated in six frames),
 * 0 = the general kernels (a hit per byte position in HBM between probe, seedextend and aggregate), 2 = the fused ones and
 * then, a read not fitting, the general ones.  Same answers either way; UMGAP_RECORDS_FUSED=0 in the environment forces 0. */
int umgap_records_last_route(void);

/* ---- the index sharded over the GPUs of one node (BASELINE.json north_star; replaces the single resident
 * `umgap prot2kmer2lca -m -o -s <socket>` server of scripts/umgap-analyse.sh:257-264) ----------------------------------
 * One rank per GPU (a process, or a thread of one process).  Rank r holds shard r of n (umgap_index_open_shard,
 * umgap_index_set_shard, umgap_synth_index_build_shard) and a replica of the taxonomy, classifies its own reads and
 * exchanges k-mers / hits with the other ranks over RCCL (xGMI): 6 bytes per k-mer out, 3 bytes per hit back (4 when a
 * value of the table needs them).  See DESIGN.md "Multi-GPU" for the pipeline and the bytes per step and peer.
 *
 * Communicator: rank 0 calls umgap_comm_unique_id and hands the 128 bytes to every rank (any channel: a file, MPI,
 * torch.distributed); every rank then calls umgap_comm_init_rank (collective).  RCCL is resolved at run time (dlopen of
 * librccl.so.1), so nothing here is needed to load the library for single-GPU use.  umgap_comm_init_local creates all
 * `world` ranks of a communicator whose exchange is device-to-device copies on ONE GPU, for rehearsing the N-rank
 * protocol where a single GPU is at hand: each rank must then be driven from its own host thread. */
typedef struct umgap_comm umgap_comm;
typedef struct umgap_sharded umgap_sharded;
int umgap_comm_unique_id(uint8_t* id128);
int umgap_comm_init_rank(const uint8_t* id128, int world, int rank, int device, umgap_comm** out);
int umgap_comm_init_local(int world, int device, umgap_comm** out_ranks /* world handles */);
/* The process-per-rank communicator WITHOUT RCCL: every rank is its own process (on its own GPU, or several sharing one --
 * which RCCL refuses), the all-to-all reads the peers' send buffers through hipIpc memory handles (device-to-device copies;
 * between GPUs they travel over xGMI peer access), the handles and the barrier live in a POSIX shared-memory block called
 * `name` ("/something": rank 0 creates it, the others wait for it, it is unlinked once all are attached).  Collective.  Each
 * exchange is host-synchronised (two barriers), so it is the rehearsal and fall-back transport, not the fast one.
 * UMGAP_IPC_TIMEOUT_S (default 120): how long a rank waits for the others before it gives up (every rank then fails). */
int umgap_comm_init_ipc(const char* name, int world, int rank, int device, umgap_comm** out);
int umgap_comm_world(const umgap_comm* c);
int umgap_comm_rank(const umgap_comm* c);
/* Collective: every rank sends a known pattern of `bytes_per_peer` bytes to every rank INCLUDING itself through the transport
 * of the sharded lookup (RCCL send/recv group, or the local hub) and checks what arrives.  With world 1 this still drives
 * ncclSend/ncclRecv (to self), so the binding can be verified on a one-GPU box. */
int umgap_comm_selftest(umgap_comm* c, size_t bytes_per_peer);
void umgap_comm_free(umgap_comm* c);
/* The sharded table of one rank: `shard` must be shard rank-of-world of the communicator; `tx` (nullable) is needed by
 * umgap_sharded_classify9_dev.  max_keys_per_call bounds one call; it is cut into `pieces` (0 = 8) pipelined pieces of
 * whole reads, so make it a multiple of pieces * kmers_per_read.  slack (0.03 if out of range): head room of the
 * fixed-size exchange regions over the mean.  Collective (the ranks agree on the value width).  Borrowed handles must
 * outlive the group. */
int umgap_sharded_create(umgap_index* shard, umgap_taxonomy* tx, umgap_comm* comm, uint64_t max_keys_per_call,
                         uint32_t pieces, double slack, umgap_sharded** out);
void umgap_sharded_free(umgap_sharded* sh);
uint32_t umgap_sharded_value_bytes(const umgap_sharded* sh);  /* 3 or 4 */
uint64_t umgap_sharded_region_slots(const umgap_sharded* sh); /* slots per peer and piece: a peer receives 6 x this out, value_bytes x this back */
uint32_t umgap_sharded_pieces(const umgap_sharded* sh);
/* How the group exchanges: 0 = store and forward (the pack fills a send buffer, the communicator moves it: RCCL send/recv groups, or
 * copies between the ranks of a local / IPC communicator); 1 = peer-written ("put"): the pack stores each key straight into the
 * OWNER's receive buffer and the owner's probe stores each value straight into the REQUESTER's gather buffer, through hipIpc mappings
 * of the peers' exchange blocks (the same mappings run over xGMI between the GPUs of a node), with per-piece counters in device memory
 * instead of copies and host barriers.  Default between the processes of an IPC communicator (UMGAP_SHARD_PUT=0: copies); opt-in
 * under RCCL and between thread-ranks that drive different devices (UMGAP_SHARD_PUT=1; thread-ranks sharing a device share its
 * hardware queues, where a spinning wait can block the kernel it waits for: they keep the copies).  In put mode umgap_sharded_free
 * is collective. */
int umgap_sharded_transport(const umgap_sharded* sh);
/* get() over the whole sharded table: as umgap_index_lookup9_dev.  Collective: every rank calls it the same number of
 * times (n may differ, 0 included); enqueues on internal streams ordered after `stream` and joins them back into it. */
int umgap_sharded_lookup9_dev(umgap_sharded* sh, const uint64_t* d_keys45, size_t n, uint32_t miss_value,
                              uint32_t* d_out_taxa, void* stream);
/* The same from / to HOST arrays (staged through the group's own device buffers; returns when out_taxa is filled).  Collective. */
int umgap_sharded_lookup9(umgap_sharded* sh, const uint64_t* keys45, size_t n, uint32_t miss_value, uint32_t* out_taxa);
/* The fused 9-mer pipeline over the sharded table: as umgap_classify9_dev / umgap_classify9_strategy_dev (any strategy),
 * results bit-identical to theirs on a whole table.  Collective like the lookup. */
int umgap_sharded_classify9_dev(umgap_sharded* sh, int method, int strategy, float factor, float lower_bound,
                                int ranked_only, const uint64_t* d_keys45, size_t n_reads, uint32_t kmers_per_read,
                                uint32_t fragments_per_read, uint32_t min_seed_size, uint32_t max_gap_size,
                                uint32_t* d_out_taxon, uint32_t* d_out_status, void* stream);
/* Waits for the device and reports (UMGAP_ERR_BUILD) whether an exchange region overflowed in any call since the last
 * check -- the affected k-mers were answered with a miss, so those calls' results are not to be used -- or (UMGAP_ERR_DEVICE,
 * put mode) whether a peer's keys or values failed to arrive within UMGAP_PUT_TIMEOUT_S (default 30 s). */
int umgap_sharded_check(umgap_sharded* sh);
/* Duration of the local probe kernel of the first piece of the last call (measurement: bench.py's roofline). */
int umgap_sharded_probe_ms(umgap_sharded* sh, float* ms);

/* Page-locked host memory for the arrays passed to the host-pointer entry points (any size; free with umgap_host_free). */
int umgap_host_alloc(size_t bytes, void** out);
void umgap_host_free(void* p);

/* ---- measurement helpers (bench.py; not part of the reference surface) ------------------- */
/* Lane-per-key probe variant of umgap_index_lookup9_dev, kept for the A/B in DESIGN.md. */
int umgap_index_lookup9_lane_dev(umgap_index* ix, const uint64_t* d_keys45, size_t n, uint32_t miss_value,
                                 uint32_t* d_out_taxa, void* stream);
/* Random-sector gather microbenchmark over the index table: each of n probes reads bytes_per_probe
 * (16..256) bytes at a bucket chosen by a hash of its index; out receives a checksum per probe.
 * bytes_per_probe | (k << 16) selects the pure-read form: k probes in flight per lane and one store per
 * lane at the end instead of one per probe (d_out still needs n entries).  Bits 24-27: footprint shift (the probes stay
 * inside the first num_buckets >> shift buckets); bits 28-30: 1 + cache-policy variant (sc0 / sc1 / nt combinations) of
 * the 64-byte, 8-in-flight form.  tools/ceiling_probe.py drives them. */
int umgap_bench_gather_dev(umgap_index* ix, size_t n, uint32_t bytes_per_probe, uint32_t* d_out, void* stream);
/* Synthetic workload generators (DESIGN.md "Synthetic workload"), all on device. */
int umgap_synth_index_build(uint64_t n_keys, const uint32_t* d_taxon_of_row, uint64_t n_tax,
                            uint64_t mul_a, uint64_t add_b, double load_factor, int device,
                            umgap_index** out);
int umgap_synth_index_build_shard(uint64_t n_keys, const uint32_t* d_taxon_of_row, uint64_t n_tax,
                                  uint64_t mul_a, uint64_t add_b, double load_factor, int device,
                                  uint32_t shard, uint32_t n_shards, umgap_index** out);
/* Synthetic tryptic-like peptides: peptide i (global number first+i) fills
 * d_bytes[d_offsets[i] .. d_offsets[i+1]) with letters determined by (seed, first+i). */
int umgap_synth_peptides_dev(uint64_t first, size_t n, uint64_t seed, const uint64_t* d_offsets,
                             uint8_t* d_bytes, int device, void* stream);
/* The same, but peptides that a tryptic digest gives back whole (no K or R inside, no leading P, the last letter K or R), and
 * with arbitrary global numbers: peptide i is number d_ids[i] (d_ids NULL: first + i).  A read for umgap_classify_tryp_dev is a
 * list of such numbers written back to back. */
int umgap_synth_tryptic_dev(const uint64_t* d_ids, uint64_t first, size_t n, uint64_t seed, const uint64_t* d_offsets,
                            uint8_t* d_bytes, int device, void* stream);
/* Synthetic proteome (surface B): an index whose keys are all 9-residue windows of the given
 * fragments (value = the fragment's taxon, smallest taxon on duplicate 9-mers), and reads made of
 * mutated copies of proteome fragments. */
int umgap_index_build_windows9_dev(const uint8_t* d_bytes, const uint64_t* d_offsets, const uint32_t* d_taxa,
                                   size_t n_fragments, uint32_t max_fragment_len, uint64_t expected_keys,
                                   double load_factor, int device, umgap_index** out);
/* The same index in WALK order (round 3; umgap_index_kind() = 10).  Consecutive 9-mers of a protein share eight residues, but a
 * hash table sends each to a bucket of its own: one 64-byte request per window.  Built from the proteome itself -- fragments in
 * the order given, as umgap_index_build_windows9_dev takes them, same values (smallest taxon on duplicate 9-mers) -- a walk array
 * lists every distinct 9-mer once, at the first window of the proteome that holds it, in proteome order: the taxon beside the low
 * key bits and a "the entry in front is the previous window of the same fragment" flag; the table maps a 9-mer to the number of
 * its entry (umgap_walk_builder_* below is the general form: values from any 9-mer table).  A run of read windows that follows the proteome is then
 * answered by one probe plus the few sectors of the walk array behind the position it returned, each step verified (exact: an
 * entry that continues a verified predecessor and carries the window's low key bits has the window's key); the windows behind a
 * break -- a changed residue, a 9-mer listed elsewhere -- are probed one by one.
 * Served by umgap_classify_prot9[_strategy]_dev (walking; umgap_index_bind_taxonomy applies) and umgap_index_lookup_prot9_dev;
 * umgap_index_lookup9[_dev], umgap_classify_records and umgap_index_dump9 take the position from the table and the taxon from the
 * walk array (same answers as the plain table, no gain); umgap_index_save / umgap_index_open keep it (UMGAPIX1 kind 10: the walk
 * array follows the stash), so every command of `umgap` runs on such a file.  Not sharded.  Up to 2^34 entries (walk_ref.hpp).
 * Measured (tools/bench_surface_b.py --walk, 2.1 G-key proteome index, profiles/r03_walk_index.json): reads that are proteome
 * fragments with 0 / 1 / 3 / 6 % of their residues changed classify 1.9 / 1.5 / 1.3 / 1.1 times as fast as over the plain table
 * (0.31 - 0.53 requests per window instead of 1.04 - 1.12); at 20 % the plain table is 1.2 times faster: a walk index is for reads
 * close to the proteome it was built from.
 * Round 4: built WITHOUT a values table beside the index under construction (entries take the taxon of the fragment that lists them; a
 * last pass over the proteome lowers each to the smallest taxon of the fragments that hold its 9-mer), so the proteome's 9-mer count
 * is bounded by what the index itself needs: 8.4 G windows build on one 288 GB GPU in 16 s (profiles/r04_walk_8g_build.json).
 * expected_keys bounds the distinct 9-mers (0: the number of windows); umgap_index_num_keys reports it as given. */
int umgap_index_build_walk9_dev(const uint8_t* d_bytes, const uint64_t* d_offsets, const uint32_t* d_taxa,
                                size_t n_fragments, uint32_t max_fragment_len, uint64_t expected_keys,
                                double load_factor, int device, umgap_index** out);
/* The general builder of a walk-order index: VALUES from any whole, unbound kind-9 table (e.g. the one umgap_index_builder_* /
 * `umgap buildindex` make from the reference's TSV, src/commands/buildindex.rs:32-48), ORDER from a proteome handed over in chunks,
 * in order (device arrays as for umgap_index_build_windows9_dev; fewer than 2^32 - 1 windows per chunk; a chunk's temporary table
 * costs 16 bytes per window of the chunk).  The walk array lists every key of the values table ONCE: at the first window of the
 * proteome that holds it, in proteome order, and the keys the proteome never showed behind them as walks of one -- 8 bytes per
 * distinct key, whatever the redundancy of the proteome; a run of at most 8 repeated 9-mers between two listed windows of a
 * fragment keeps entries of its own (nothing points at them: they keep a walk through the fragment in one piece), as long as these
 * stay within an eighth of the keys; windows of the proteome whose 9-mer the values table does not hold get no entry (a miss stays
 * a miss).  Every answer is that of the values table.  Up to 2^34 entries (round 4: from 2^32 - 2^22 entries on the table holds the
 * entry number >> 1 or >> 2 and the key's low bits pick the entry of the group, umgap_amd/csrc/walk_ref.hpp).  `values` must outlive
 * the builder (it is only read); finish consumes the builder. */
typedef struct umgap_walk_builder umgap_walk_builder;
int umgap_walk_builder_new(const umgap_index* values, double load_factor, umgap_walk_builder** out);
int umgap_walk_builder_add_dev(umgap_walk_builder* b, const uint8_t* d_bytes, const uint64_t* d_offsets, size_t n_fragments,
                               uint32_t max_fragment_len);
int umgap_walk_builder_add(umgap_walk_builder* b, const uint8_t* bytes, const uint64_t* offsets, size_t n_fragments,
                           uint32_t max_fragment_len);   /* the same from host memory */
int umgap_walk_builder_finish(umgap_walk_builder* b, umgap_index** out);
/* The same WITHOUT a values table in HBM (round 5): the order first, the values afterwards.  The general builder above keeps three
 * things resident -- the values table, the position table under construction and the walk array: 6.5 G keys on one 288 GB GPU -- so
 * the real 9-mer index (8 G keys) with the reference's own values (`printindex | umgap buildindex --proteome`) did not fit.  Here
 *   1. umgap_walk_builder_new_deferred(expected_keys = an upper bound on the distinct 9-mers of the proteome),
 *   2. umgap_walk_builder_add[_dev] for every chunk of the proteome: EVERY 9-mer of the proteome gets an entry, valued 0 = "no key",
 *   3. umgap_walk_builder_set_values[_dev](keys45, values) for the (key, value) pairs of the index, in chunks of any size and order
 *      (values > 0; a key twice with the same value is fine, with two values an error at finish): a key the proteome showed has its
 *      entry valued; the others are collected (room for expected_keys / 16 of them) and appended behind the proteome as walks of one,
 *   4. umgap_walk_builder_finish.
 * Resident: the position table and the walk array only (204 GB for 8.4 G proteome 9-mers) plus one chunk.  Answers: a key's value;
 * for everything else -- 9-mers of the proteome that were given no value included -- the miss value, as from the plain table of the
 * same (key, value) pairs; a walk runs THROUGH such a 9-mer (its entry says "no key") instead of breaking at it.
 * umgap_index_num_keys counts the valued keys; umgap_index_dump9 lists only those; the file keeps the flag (UMGAPIX1 kind 10,
 * bit 62 of the walk-array word of the header). */
int umgap_walk_builder_new_deferred(uint64_t expected_keys, double load_factor, int device, umgap_walk_builder** out);
int umgap_walk_builder_set_values_dev(umgap_walk_builder* b, const uint64_t* d_keys45, const uint32_t* d_values, size_t n);
int umgap_walk_builder_set_values(umgap_walk_builder* b, const uint64_t* keys45, const uint32_t* values, size_t n);
/* How a walk-order index numbers its entries: the table holds entry number >> *shift (0 below 2^32 - 2^22 entries, else 1 or 2), the
 * walk array has *n_entries entries (padding included), and *n_exceptions keys share their low 31 bits with another key of their
 * group of 2^shift entries (looked up through a sorted list).  Any pointer may be null. */
int umgap_index_walk_info(const umgap_index* ix, uint32_t* shift, uint64_t* n_entries, uint32_t* n_exceptions);
void umgap_walk_builder_free(umgap_walk_builder* b);
int umgap_synth_prot_reads_dev(const uint8_t* d_proteome, uint64_t n_proteome_fragments, uint32_t fragment_len,
                               uint64_t seed, uint64_t first_read, size_t n_reads, uint32_t fragments_per_read,
                               uint32_t mutations_per_mille, uint8_t* d_out, int device, void* stream);
/* Reads of the benchmark: per pair, path_percent of the k-mers are members whose taxon lies on the root path of the pair's
 * source taxon, unrelated_percent are members with unrelated taxa (slot pairs share one), the rest are non-members (1 % of
 * all k-mers carry an 'X').  umgap_synth_reads9_dev is the 60 / 10 mix of the headline; bench.py's workload_sweep varies it. */
int umgap_synth_reads9_mix_dev(const umgap_taxonomy* tx, const uint32_t* d_taxon_of_row, const uint32_t* d_row_of_id,
                               uint64_t n_tax, uint64_t n_keys, uint64_t mul_a, uint64_t add_b, uint64_t seed,
                               uint64_t first_pair, uint64_t n_pairs, uint32_t kmers_per_pair, uint32_t path_percent,
                               uint32_t unrelated_percent, uint64_t* d_keys45, void* stream);
int umgap_synth_reads9_dev(const umgap_taxonomy* tx, const uint32_t* d_taxon_of_row,
                           const uint32_t* d_row_of_id, uint64_t n_tax, uint64_t n_keys, uint64_t mul_a,
                           uint64_t add_b, uint64_t seed, uint64_t first_pair, uint64_t n_pairs,
                           uint32_t kmers_per_pair, uint64_t* d_keys45, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UMGAP_HIP_H */
