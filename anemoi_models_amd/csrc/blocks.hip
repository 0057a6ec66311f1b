// Block-level entry points of the C ABI: one call issues a whole GraphTransformer block's launch sequence on the caller's
// stream (include/anemoi_amd.h, "Block-level entry points").  Nothing here touches the device itself: the functions
// validate the argument block and call the op-level entry points in the order the reference's block runs them
// (layers/block.py:602-635).
#include "common.hpp"

using namespace anemoi;

namespace {

int check_args(const anemoi_gt_block_args* a, const char* who, bool with_input) {
  ANEMOI_REQUIRE(a != nullptr, ANEMOI_ERR_INVALID, "%s: null argument block", who);
  ANEMOI_REQUIRE(a->struct_bytes == (int64_t)sizeof(anemoi_gt_block_args), ANEMOI_ERR_INVALID,
                 "%s: argument block of %lld bytes, this library expects %lld (header / binding out of sync)", who,
                 (long long)a->struct_bytes, (long long)sizeof(anemoi_gt_block_args));
  ANEMOI_REQUIRE(a->dtype == ANEMOI_BF16, ANEMOI_ERR_UNSUPPORTED,
                 "%s: the LayerNorm-folded block route is bf16 only (dtype %d)", who, a->dtype);
  ANEMOI_REQUIRE(a->n_dst >= 0 && a->C > 0 && a->H > 0 && a->C % a->H == 0 && a->up > 0 && a->hidden > 0,
                 ANEMOI_ERR_INVALID, "%s: bad shape (n_dst %lld, C %d, H %d, up %d, hidden %d)", who, (long long)a->n_dst,
                 a->C, a->H, a->up, a->hidden);
  ANEMOI_REQUIRE(a->k_proj >= a->C + a->H * a->up && a->ld_att >= a->k_proj, ANEMOI_ERR_INVALID,
                 "%s: projection K %d / ld_att %lld too small for C + H * up = %d", who, a->k_proj, (long long)a->ld_att,
                 a->C + a->H * a->up);
  ANEMOI_REQUIRE(a->edge_attr && a->rowptr && a->col && a->att && a->w_proj && a->y && a->y_stats && a->w_fc1 && a->cs_fc1 &&
                     a->h && a->w_fc2 && a->out && a->stats_ws,
                 ANEMOI_ERR_INVALID, "%s: null pointer", who);
  ANEMOI_REQUIRE(a->stats_ws_bytes >= a->n_dst * (int64_t)(a->C / 128 > 1 ? a->C / 128 : 1) * 8, ANEMOI_ERR_INVALID,
                 "%s: statistics workspace of %lld bytes too small", who, (long long)a->stats_ws_bytes);
  if (with_input) {
    ANEMOI_REQUIRE(a->x && a->x_stats && a->w_in && a->cs_in && a->sq, ANEMOI_ERR_INVALID, "%s: null pointer (input product)",
                   who);
    ANEMOI_REQUIRE(a->n_in >= 4 * a->C + a->H * a->up && a->ld_sq >= a->n_in && a->ldx >= a->C, ANEMOI_ERR_INVALID,
                   "%s: input product of %d columns (ld %lld) cannot hold x_r | q | k | v | u = %d", who, a->n_in,
                   (long long)a->ld_sq, 4 * a->C + a->H * a->up);
  } else {
    ANEMOI_REQUIRE(a->q && a->k && a->v && a->x_r && a->u && a->res, ANEMOI_ERR_INVALID,
                   "%s: null pointer (edge phase operands / residual)", who);
  }
  return ANEMOI_OK;
}

int run_tail(const anemoi_gt_block_args* a, anemoi_stream_t stream) {
  int st;
  if (a->tile_hdr != nullptr && a->run_ptr == nullptr)
    st = anemoi_gt_edge_attention_folded_tiles(a->dtype, a->q, a->ldq, a->k, a->v, a->ldkv, a->x_r, a->ldr, a->u, a->ldu,
                                               a->edge_attr, a->up, a->rowptr, a->col, a->tile_hdr, a->tile_dst, a->tile_src,
                                               a->tile_slot, a->tile_xcd, a->tile_max_per_xcd, a->tile_src_cap,
                                               a->tile_edge_cap, a->n_src, a->n_edges, a->att, a->ld_att, nullptr, a->n_dst,
                                               a->C, a->H, stream);
  else if (a->sched != nullptr && a->run_ptr == nullptr)
    st = anemoi_gt_edge_attention_folded_sched(a->dtype, a->q, a->ldq, a->k, a->v, a->ldkv, a->x_r, a->ldr, a->u, a->ldu,
                                               a->edge_attr, a->up, a->rowptr, a->col, a->sched, a->sched_slots, a->sched_steps,
                                               a->n_src, a->n_edges, a->att, a->ld_att, nullptr, a->n_dst, a->C, a->H, stream);
  else if (a->run_ptr != nullptr && a->run_dst != nullptr)
    st = anemoi_gt_edge_attention_folded_groups(a->dtype, a->q, a->ldq, a->k, a->v, a->ldkv, a->x_r, a->ldr, a->u, a->ldu,
                                                a->edge_attr, a->up, a->rowptr, a->col, a->run_ptr, a->run_dst, a->run_perm,
                                                a->n_runs, a->n_src, a->att, a->ld_att, nullptr, a->n_dst, a->C, a->H, stream);
  else
    st = anemoi_gt_edge_attention_folded_runs(a->dtype, a->q, a->ldq, a->k, a->v, a->ldkv, a->x_r, a->ldr, a->u, a->ldu,
                                              a->edge_attr, a->up, a->rowptr, a->col, a->run_ptr, a->run_perm, a->n_runs,
                                              a->att, a->ld_att, nullptr, a->n_dst, a->C, a->H, stream);
  if (st != ANEMOI_OK) return st;
  // y = projection(att) + res, with { rstd, -mean rstd } of y's rows for the node MLP's LayerNorm
  st = anemoi_linear_stats(a->dtype, a->att, a->ld_att, a->w_proj, a->b_proj, nullptr, nullptr, a->res, a->ld_res, a->y,
                           a->C, a->n_dst, a->C, a->k_proj, a->stats_ws, a->stats_ws_bytes, a->eps_mlp, a->y_stats, stream);
  if (st != ANEMOI_OK) return st;
  // h = act(Linear(LayerNorm(y)))
  st = anemoi_linear_ln(a->dtype, a->dtype, a->y, a->C, a->w_fc1, a->b_fc1, a->cs_fc1, a->y_stats, nullptr, 0, a->h,
                        a->hidden, a->n_dst, a->hidden, a->C, a->act, stream);
  if (st != ANEMOI_OK) return st;
  // out = Linear(h) + y (+ the statistics of the LayerNorm that reads out next)
  if (a->out_stats != nullptr)
    return anemoi_linear_stats(a->dtype, a->h, a->hidden, a->w_fc2, a->b_fc2, nullptr, nullptr, a->y, a->C, a->out, a->C,
                               a->n_dst, a->C, a->hidden, a->stats_ws, a->stats_ws_bytes, a->eps_out, a->out_stats, stream);
  return anemoi_linear(a->dtype, a->dtype, a->h, a->hidden, a->w_fc2, a->b_fc2, a->y, a->C, a->out, a->C, a->n_dst, a->C,
                       a->hidden, ANEMOI_ACT_NONE, stream);
}

}  // namespace

extern "C" int anemoi_gt_block_tail(const anemoi_gt_block_args* args, anemoi_stream_t stream) {
  const int st = check_args(args, "anemoi_gt_block_tail", false);
  if (st != ANEMOI_OK) return st;
  if (args->n_dst == 0) return ANEMOI_OK;
  return run_tail(args, stream);
}

extern "C" int anemoi_gt_processor_block_forward(const anemoi_gt_block_args* args, anemoi_stream_t stream) {
  int st = check_args(args, "anemoi_gt_processor_block_forward", true);
  if (st != ANEMOI_OK) return st;
  if (args->n_dst == 0) return ANEMOI_OK;
  // sq = Linear'(x) with LayerNorm(x) folded in: [x_r | q | k | v | u]
  st = anemoi_linear_ln(args->dtype, args->dtype, args->x, args->ldx, args->w_in, args->b_in, args->cs_in, args->x_stats,
                        nullptr, 0, args->sq, args->ld_sq, args->n_dst, args->n_in, args->C, ANEMOI_ACT_NONE, stream);
  if (st != ANEMOI_OK) return st;
  anemoi_gt_block_args a = *args;
  const char* sq = static_cast<const char*>(args->sq);
  const int64_t c2 = (int64_t)args->C * 2;  // bytes of C bf16 columns
  a.x_r = sq;
  a.q = sq + c2;
  a.k = sq + 2 * c2;
  a.v = sq + 3 * c2;
  a.u = sq + 4 * c2;
  a.ldq = a.ldkv = a.ldr = a.ldu = args->ld_sq;
  a.res = args->x;
  a.ld_res = args->ldx;
  return run_tail(&a, stream);
}

extern "C" int anemoi_transformer_block_forward(const anemoi_tfm_block_args* a, anemoi_stream_t stream) {
  const char* who = "anemoi_transformer_block_forward";
  ANEMOI_REQUIRE(a != nullptr, ANEMOI_ERR_INVALID, "%s: null argument block", who);
  ANEMOI_REQUIRE(a->struct_bytes == (int64_t)sizeof(anemoi_tfm_block_args), ANEMOI_ERR_INVALID,
                 "%s: argument block of %lld bytes, this library expects %lld (header / binding out of sync)", who,
                 (long long)a->struct_bytes, (long long)sizeof(anemoi_tfm_block_args));
  ANEMOI_REQUIRE(a->dtype == ANEMOI_F32 || a->dtype == ANEMOI_BF16, ANEMOI_ERR_UNSUPPORTED, "%s: dtype %d", who, a->dtype);
  ANEMOI_REQUIRE(a->rows >= 0 && a->B > 0 && a->S >= 0 && a->rows == (int64_t)a->B * a->S && a->C > 0 && a->H > 0 &&
                     a->C % a->H == 0 && a->hidden > 0,
                 ANEMOI_ERR_INVALID, "%s: bad shape (rows %lld, B %d, S %d, C %d, H %d, hidden %d)", who, (long long)a->rows,
                 a->B, a->S, a->C, a->H, a->hidden);
  ANEMOI_REQUIRE(a->x && a->ln1_w && a->ln1_b && a->ln2_w && a->ln2_b && a->w_qkv && a->w_proj && a->w_fc1 && a->w_fc2 &&
                     a->h_ln && a->qkv && a->att && a->y && a->h && a->out,
                 ANEMOI_ERR_INVALID, "%s: null pointer", who);
  ANEMOI_REQUIRE(a->ldx >= a->C && a->out != a->x, ANEMOI_ERR_INVALID, "%s: ldx too small, or out aliases x", who);
  if (a->rows == 0) return ANEMOI_OK;
  const int C = a->C;
  int st = anemoi_layer_norm(a->dtype, a->x, a->ldx, a->ln1_w, a->ln1_b, a->h_ln, C, a->rows, C, a->eps1, stream);
  if (st != ANEMOI_OK) return st;
  st = anemoi_linear(a->dtype, a->dtype, a->h_ln, C, a->w_qkv, a->b_qkv, nullptr, 0, a->qkv, 3 * (int64_t)C, a->rows, 3 * C, C,
                     ANEMOI_ACT_NONE, stream);
  if (st != ANEMOI_OK) return st;
  st = anemoi_mhsa(a->dtype, a->qkv, 3 * (int64_t)C, a->att, C, a->mhsa_ws, nullptr, a->B, a->S, a->H, C / a->H, a->window,
                   a->dropout_p, a->dropout_seed, a->dropout_seed_dev, a->dropout_h0, a->dropout_h_total, stream);
  if (st != ANEMOI_OK) return st;
  st = anemoi_linear(a->dtype, a->dtype, a->att, C, a->w_proj, a->b_proj, a->x, a->ldx, a->y, C, a->rows, C, C, ANEMOI_ACT_NONE,
                     stream);  // x + attention(...)
  if (st != ANEMOI_OK) return st;
  st = anemoi_layer_norm(a->dtype, a->y, C, a->ln2_w, a->ln2_b, a->h_ln, C, a->rows, C, a->eps2, stream);
  if (st != ANEMOI_OK) return st;
  st = anemoi_linear(a->dtype, a->dtype, a->h_ln, C, a->w_fc1, a->b_fc1, nullptr, 0, a->h, a->hidden, a->rows, a->hidden, C,
                     a->act, stream);
  if (st != ANEMOI_OK) return st;
  return anemoi_linear(a->dtype, a->dtype, a->h, a->hidden, a->w_fc2, a->b_fc2, a->y, C, a->out, C, a->rows, C, a->hidden,
                       ANEMOI_ACT_NONE, stream);  // y + mlp(...)
}
