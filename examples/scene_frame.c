/* A C host for the HIP scene path: what the reference's render thread would do
 * through include/vkr_hip_scene.h, with no Python and no C++ in between.
 *
 *   scene_frame <scene.bin> <result.bin> [payload.bin]
 *
 * scene.bin  (little endian): uint64 nodes, levels, rows, views; then
 *            parent[nodes] u32, level_offsets[levels+1] u64, pos/quat/scale
 *            [nodes][4] f32 each, node[rows] u32, ident[rows][8] u32,
 *            matbits[rows] u32, bounds[rows][12] f32, planes[views][24] f32,
 *            required[views] u32, camera view[16] f32.
 * result.bin: vkrhip_frame_info, world[nodes][16] f32, then per view its visible
 *            list (visible_count u32), then the sorted draw keys (u64) and the
 *            VkrDrawItem rows (48 B each).
 * payload.bin (when named): the frame also produces the candidate rows and the
 *            file takes what vkrhip_fill_world_payload hands back, i.e. the
 *            reference's VkrWorldPassPayload: the 88-byte struct itself (its
 *            pointers are meaningless in a file), the 12-byte VkrVisibilityStats,
 *            then the arrays it points to: candidate rows (144 B), transmission
 *            rows (144 B), draws (48 B), instances (80 B).
 * Exit code 0 on success; any library error is printed with its message. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "vkr_hip_scene.h"

static void *read_array(FILE *f, size_t count, size_t size) {
  void *p = malloc(count * size + 1);
  if (!p || fread(p, size, count, f) != count) {
    fprintf(stderr, "scene file is short\n");
    exit(2);
  }
  return p;
}

#define CHECK(ctx, call)                                                         \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != VKRHIP_OK) {                                                      \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, vkrhip_last_error(ctx));     \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(int argc, char **argv) {
  if (argc != 3 && argc != 4) {
    fprintf(stderr, "usage: %s scene.bin result.bin [payload.bin]\n", argv[0]);
    return 2;
  }
  FILE *in = fopen(argv[1], "rb");
  if (!in) {
    perror(argv[1]);
    return 2;
  }
  uint64_t head[4];
  if (fread(head, 8, 4, in) != 4) return 2;
  const uint64_t nodes = head[0], levels = head[1], rows = head[2], views = head[3];
  uint32_t *parent = read_array(in, nodes, 4);
  uint64_t *level_offsets = read_array(in, levels + 1, 8);
  float *pos = read_array(in, nodes * 4, 4);
  float *quat = read_array(in, nodes * 4, 4);
  float *scale = read_array(in, nodes * 4, 4);
  uint32_t *node = read_array(in, rows, 4);
  uint32_t *ident = read_array(in, rows * 8, 4);
  uint32_t *matbits = read_array(in, rows, 4);
  float *bounds = read_array(in, rows * 12, 4);
  float *planes = read_array(in, views * 24, 4);
  uint32_t *required = read_array(in, views, 4);
  float *camera_view = read_array(in, 16, 4);
  fclose(in);

  vkrhip_config cfg = {0};
  cfg.struct_size = sizeof cfg;
  cfg.max_nodes = nodes;
  cfg.max_candidates = rows;
  cfg.max_views = (uint32_t)views;
  vkrhip_ctx *ctx = NULL;
  if (vkrhip_create(&cfg, &ctx) != VKRHIP_OK) {
    fprintf(stderr, "vkrhip_create: %s\n", vkrhip_last_error(NULL));
    return 1;
  }
  CHECK(ctx, vkrhip_set_hierarchy(ctx, parent, nodes, level_offsets, (uint32_t)levels));
  CHECK(ctx, vkrhip_upload_trs(ctx, 0, nodes, pos, quat, scale));
  /* row i drawn with transform i for every row: the identity mapping is passed as
     NULL (the library then fuses transform and cull) */
  int identity = rows == nodes;
  for (uint64_t i = 0; identity && i < rows; ++i) identity = node[i] == i;
  CHECK(ctx, vkrhip_set_candidates(ctx, rows, identity ? NULL : node, ident, matbits, bounds));
  CHECK(ctx, vkrhip_set_views(ctx, (uint32_t)views, planes, required, camera_view));
  CHECK(ctx, vkrhip_frame(ctx, VKRHIP_STAGE_ALL | (argc == 4 ? VKRHIP_STAGE_CANDIDATES : 0u)));
  if (argc == 4) { /* what replaces application_build_world_payload (application.h:671-961) */
    vkrhip_world_payload payload;
    vkrhip_visibility_stats stats;
    CHECK(ctx, vkrhip_fill_world_payload(ctx, &payload, &stats));
    FILE *pf = fopen(argv[3], "wb");
    if (!pf) {
      perror(argv[3]);
      return 2;
    }
    fwrite(&payload, sizeof payload, 1, pf);
    fwrite(&stats, sizeof stats, 1, pf);
    fwrite(payload.gpu_candidates, 144, payload.gpu_candidate_count, pf);
    fwrite(payload.transmission_gpu_candidates, 144, payload.transmission_gpu_candidate_count, pf);
    fwrite(payload.transparent_draws, 48, payload.transparent_draw_count, pf);
    fwrite(payload.instances, 80, payload.instance_count, pf);
    fclose(pf);
  }
  vkrhip_frame_info info;
  CHECK(ctx, vkrhip_get_frame_info(ctx, &info));

  FILE *out = fopen(argv[2], "wb");
  if (!out) {
    perror(argv[2]);
    return 2;
  }
  fwrite(&info, sizeof info, 1, out);
  float *world = malloc(nodes * 64 + 64);
  CHECK(ctx, vkrhip_read_world(ctx, 0, nodes, world));
  fwrite(world, 64, nodes, out);
  for (uint32_t v = 0; v < views; ++v) {
    const uint32_t count = info.views[v].visible_count;
    uint32_t *list = malloc((size_t)count * 4 + 4);
    if (count) CHECK(ctx, vkrhip_read_visible(ctx, v, count, list));
    fwrite(list, 4, count, out);
    free(list);
  }
  const uint32_t draws = info.transparent_draw_count;
  uint64_t *keys = malloc((size_t)draws * 8 + 8);
  void *items = malloc((size_t)draws * 48 + 48);
  if (draws) CHECK(ctx, vkrhip_read_transparent(ctx, draws, items, NULL, keys, NULL));
  fwrite(keys, 8, draws, out);
  fwrite(items, 48, draws, out);
  fclose(out);
  printf("nodes %llu rows %llu views %llu: updated %u, visible[0] %u, transparent draws %u\n",
         (unsigned long long)nodes, (unsigned long long)rows, (unsigned long long)views,
         info.nodes_updated, info.views[0].visible_count, draws);
  vkrhip_destroy(ctx);
  return 0;
}
