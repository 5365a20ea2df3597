/* vus_oracle.h -- declarations private to the CPU oracle / CPU port (TEST INFRASTRUCTURE ONLY).
 *
 * vus_ba_structure: the co-observation PAIR LISTS of the reduced camera system -- lower block band of half-width `band`
 * pose blocks; non-zero block (blk_i >= blk_k) number q owns pairs [blk_ptr[q], blk_ptr[q+1]); pair p = (pair_a: P-order
 * slot of pose blk_i, pair_b: P-order slot of pose blk_k) seeing the same point.  Rounds 1-3 of the HIP library walked
 * these lists (include/vus.h had the struct); round 4 replaced them by the tile pairs of vus_ba_tiles.  The multi-threaded
 * CPU port (vus_oracle_ba_mt.c, the cpu_baseline of bench.py) still eliminates the landmarks block by block over them,
 * with its own W / Y in P-order. */
#ifndef VUS_ORACLE_H
#define VUS_ORACLE_H
#include "../include/vus.h"

typedef struct vus_ba_structure {
  int band;
  int n_blocks;
  int n_pairs;
  const int* blk_ptr;      /* [n_blocks+1] */
  const int* blk_i;        /* [n_blocks] */
  const int* blk_k;        /* [n_blocks] */
  const int* pair_a;       /* [n_pairs] */
  const int* pair_b;       /* [n_pairs] */
} vus_ba_structure;

#endif
