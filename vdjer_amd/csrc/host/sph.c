/* sph.c -- see sph.h */
#include "sph.h"

#include <stdlib.h>
#include <string.h>

uint64_t sph_murmur64a(const void* key, int len, uint64_t seed) {
	const uint64_t m = 0xc6a4a7935bd1e995ULL;
	const int r = 47;
	uint64_t h = seed ^ ((uint64_t) len * m);
	const unsigned char* p = (const unsigned char*) key;
	int nblocks = len / 8;
	for (int i = 0; i < nblocks; i++) {
		uint64_t k;
		memcpy(&k, p + 8 * i, 8);
		k *= m; k ^= k >> r; k *= m;
		h ^= k; h *= m;
	}
	const unsigned char* t = p + 8 * nblocks;
	switch (len & 7) {
	case 7: h ^= (uint64_t) t[6] << 48; /* fallthrough */
	case 6: h ^= (uint64_t) t[5] << 40; /* fallthrough */
	case 5: h ^= (uint64_t) t[4] << 32; /* fallthrough */
	case 4: h ^= (uint64_t) t[3] << 24; /* fallthrough */
	case 3: h ^= (uint64_t) t[2] << 16; /* fallthrough */
	case 2: h ^= (uint64_t) t[1] << 8;  /* fallthrough */
	case 1: h ^= (uint64_t) t[0]; h *= m;
	}
	h ^= h >> r; h *= m; h ^= h >> r;
	return h;
}

#define MIN_BUCKETS 4
#define START_BUCKETS 32

static size_t t_hash(const sph_table* t, const char* key) {
	int len = t->keylen > 0 ? t->keylen : (int) strlen(key);
	return (size_t) (sph_murmur64a(key, len, 97) / sizeof(void*));      /* hash_munger<HashKey*> */
}

static int t_eq(const sph_table* t, const char* a, const char* b) {
	if (a == b) return 1;
	return t->keylen > 0 ? strncmp(a, b, (size_t) t->keylen) == 0 : strcmp(a, b) == 0;
}

static size_t enlarge_threshold(const sph_table* t, size_t n) { return (size_t) ((float) n * t->enlarge); }
static size_t shrink_threshold(const sph_table* t, size_t n) { return (size_t) ((float) n * t->shrink); }

/* hashtable-common.h:329-343 */
static size_t min_buckets(const sph_table* t, size_t num_elts, size_t wanted) {
	size_t sz = MIN_BUCKETS;
	while (sz < wanted || num_elts >= (size_t) ((float) sz * t->enlarge)) sz *= 2;
	return sz;
}

void sph_init(sph_table* t, int keylen, int sparse) {
	memset(t, 0, sizeof *t);
	t->keylen = keylen;
	t->enlarge = sparse ? 0.8f : 0.5f;
	t->shrink = sparse ? 0.8f * 0.4f : 0.5f * 0.4f;
	t->nbuckets = START_BUCKETS;
	t->b = (sph_bucket*) calloc(t->nbuckets, sizeof(sph_bucket));
}

void sph_free(sph_table* t) {
	free(t->b);
	t->b = NULL;
}

size_t sph_size(const sph_table* t) { return t->num_elements - t->num_deleted; }

size_t sph_next(const sph_table* t, size_t i) {
	while (i < t->nbuckets && (!t->b[i].key || t->b[i].deleted)) i++;
	return i;
}

/* copy_from (densehashtable.h:631-653): live entries in old bucket order, probing for an empty bucket */
static void rehash(sph_table* t, size_t min_wanted) {
	size_t live = sph_size(t);
	size_t nn = min_buckets(t, live, min_wanted);
	sph_bucket* nb = (sph_bucket*) calloc(nn, sizeof(sph_bucket));
	for (size_t i = 0; i < t->nbuckets; i++) {
		if (!t->b[i].key || t->b[i].deleted) continue;
		size_t probes = 0, bk = t_hash(t, t->b[i].key) & (nn - 1);
		while (nb[bk].key) { probes++; bk = (bk + probes) & (nn - 1); }
		nb[bk] = t->b[i];
	}
	free(t->b);
	t->b = nb;
	t->nbuckets = nn;
	t->num_elements = live;
	t->num_deleted = 0;
	t->consider_shrink = 0;
}

/* densehashtable.h:539-566 */
static int maybe_shrink(sph_table* t) {
	int ret = 0;
	size_t remain = sph_size(t);
	size_t thr = shrink_threshold(t, t->nbuckets);
	if (thr > 0 && remain < thr && t->nbuckets > START_BUCKETS) {
		size_t sz = t->nbuckets / 2;
		while (sz > START_BUCKETS && (float) remain < (float) sz * t->shrink) sz /= 2;   /* float compare as in the library */
		rehash(t, sz);
		ret = 1;
	}
	t->consider_shrink = 0;
	return ret;
}

/* densehashtable.h:571-616 */
static int resize_delta(sph_table* t, size_t delta) {
	int did = 0;
	if (t->consider_shrink && maybe_shrink(t)) did = 1;
	if (t->nbuckets >= MIN_BUCKETS && t->num_elements + delta <= enlarge_threshold(t, t->nbuckets)) return did;
	size_t needed = min_buckets(t, t->num_elements + delta, 0);
	if (needed <= t->nbuckets) return did;
	size_t resize_to = min_buckets(t, t->num_elements - t->num_deleted + delta, t->nbuckets);
	if (resize_to < needed) {
		size_t target = shrink_threshold(t, resize_to * 2);
		if (t->num_elements - t->num_deleted + delta >= target) resize_to *= 2;
	}
	rehash(t, resize_to);
	return 1;
}

/* find_position (densehashtable.h:824-848): *insert_pos = first deleted bucket seen, else the empty bucket */
static size_t find_position(const sph_table* t, const char* key, size_t* insert_pos) {
	size_t probes = 0, mask = t->nbuckets - 1;
	size_t bk = t_hash(t, key) & mask;
	size_t ins = (size_t) -1;
	for (;;) {
		const sph_bucket* b = &t->b[bk];
		if (!b->key) {
			if (insert_pos) *insert_pos = ins == (size_t) -1 ? bk : ins;
			return (size_t) -1;
		} else if (b->deleted) {
			if (ins == (size_t) -1) ins = bk;
		} else if (t_eq(t, key, b->key)) {
			return bk;
		}
		probes++;
		bk = (bk + probes) & mask;
	}
}

size_t sph_find(const sph_table* t, const char* key) {
	if (sph_size(t) == 0) return (size_t) -1;
	return find_position(t, key, NULL);
}

static size_t insert_at(sph_table* t, size_t pos, const char* key, void* val) {
	if (t->b[pos].key && t->b[pos].deleted) t->num_deleted--;
	else t->num_elements++;
	t->b[pos].key = key;
	t->b[pos].val = val;
	t->b[pos].deleted = 0;
	return pos;
}

size_t sph_map_put(sph_table* t, const char* key, void* val, int* inserted) {
	size_t ins;
	size_t pos = find_position(t, key, &ins);
	if (pos != (size_t) -1) {
		t->b[pos].val = val;
		if (inserted) *inserted = 0;
		return pos;
	}
	if (inserted) *inserted = 1;
	if (resize_delta(t, 1)) {
		pos = find_position(t, key, &ins);
		return insert_at(t, ins, key, val);
	}
	return insert_at(t, ins, key, val);
}

size_t sph_set_insert(sph_table* t, const char* key, int* inserted) {
	size_t ins;
	resize_delta(t, 1);
	size_t pos = find_position(t, key, &ins);
	if (pos != (size_t) -1) {
		if (inserted) *inserted = 0;
		return pos;
	}
	if (inserted) *inserted = 1;
	return insert_at(t, ins, key, NULL);
}

void sph_erase_at(sph_table* t, size_t bucket) {
	if (bucket >= t->nbuckets || !t->b[bucket].key || t->b[bucket].deleted) return;
	t->b[bucket].deleted = 1;
	t->num_deleted++;
	t->consider_shrink = 1;
}

int sph_erase(sph_table* t, const char* key) {
	size_t pos = sph_find(t, key);
	if (pos == (size_t) -1) return 0;
	sph_erase_at(t, pos);
	return 1;
}

void sph_resize0(sph_table* t) { maybe_shrink(t); }
