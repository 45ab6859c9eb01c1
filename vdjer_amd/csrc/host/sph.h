/*
 * sph.h -- iteration-order-exact emulation of the google-sparsehash 2.x containers the reference uses.
 *
 * The reference's outputs follow hash-table iteration order: contig order in
 * vdj_contigs.fa, which overlapping window is dropped, root order, per-contig window order.  These tables
 * reproduce that order without the library: same hash (MurmurHash64A seed 97, divided by 8 for pointer
 * keys, hashtable-common.h:352-361), same triangular probing (densehashtable.h:119,824-848), same growth
 * and shrink rules (densehashtable.h:539-616, 631-653; sparsehashtable.h same with occupancy 0.8), same
 * "first deleted bucket seen" insert position, iteration = ascending bucket index.
 * Pinned by tests/test_host_cpu.py against dumps of the real library (tests/golden/order_*, *.node_order.*).
 */
#ifndef VDJ_SPH_H
#define VDJ_SPH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

uint64_t sph_murmur64a(const void* key, int len, uint64_t seed);   /* hash_utils.c:5-46 */

typedef struct {
	const char* key;      /* NULL = empty bucket */
	void* val;
	uint8_t deleted;
} sph_bucket;

typedef struct {
	sph_bucket* b;
	size_t nbuckets, num_elements /* occupied incl. deleted */, num_deleted;
	int keylen;           /* > 0: keys compare/hash over exactly keylen chars (my_hash/contig_hash); 0: strlen (vjf_hash) */
	float enlarge, shrink;   /* 0.5/0.2 dense, 0.8/0.32 sparse */
	int consider_shrink;
} sph_table;

void sph_init(sph_table* t, int keylen, int sparse);
void sph_free(sph_table* t);
size_t sph_size(const sph_table* t);
/* bucket index of the key or (size_t)-1 */
size_t sph_find(const sph_table* t, const char* key);
/* dense_hash_map::operator[] = find_or_insert (densehashtable.h:982-998): returns the bucket; *inserted tells */
size_t sph_map_put(sph_table* t, const char* key, void* val, int* inserted);
/* dense/sparse_hash_set::insert (resize_delta first, densehashtable.h:966-969) */
size_t sph_set_insert(sph_table* t, const char* key, int* inserted);
void sph_erase_at(sph_table* t, size_t bucket);                     /* erase(iterator), densehashtable.h:1024-1030 */
int sph_erase(sph_table* t, const char* key);
void sph_resize0(sph_table* t);                                     /* resize(0), densehashtable.h:660-665 */
/* iteration: first live bucket at or after i, or nbuckets */
size_t sph_next(const sph_table* t, size_t i);

#ifdef __cplusplus
}
#endif
#endif
