/* Compiled by tests/test_host_cabi.py with `gcc -std=c99 -Wall -Werror -Iinclude`: the public header is
 * plain C (no HIP types) and a C host links the library directly.  Calls only entry points that need no GPU. */
#include <stdio.h>
#include <string.h>
#include "immoco_hip.h"

int main(void) {
  immoco_grid_cfg cfg;
  immoco_grid_geometry geo;
  char msg[256];
  int l;
  if (immoco_version() < 100) return 1;
  memset(&cfg, 0, sizeof cfg);
  cfg.dims = 3;
  cfg.n_levels = 16;
  cfg.n_features = 2;
  cfg.log2_hashmap_size = 19;
  cfg.base_resolution = 16;
  cfg.per_level_scale = 2.0f;
  if (immoco_grid_geometry_query(&cfg, &geo) != IMMOCO_OK) return 2;
  printf("entries %u", (unsigned)geo.offset[16]);
  for (l = 0; l < 16; ++l) printf(" %u:%d", (unsigned)geo.resolution[l], (int)geo.hashed[l]);
  printf("\n");
  cfg.dims = 5; /* invalid: status code + message, no exception crosses the ABI */
  if (immoco_grid_geometry_query(&cfg, &geo) == IMMOCO_OK) return 3;
  immoco_last_error(msg, sizeof msg);
  if (strlen(msg) == 0) return 4;
  return 0;
}
