/* TEST INFRASTRUCTURE - never shipped, never linked into the product. A stand-in for librocoder_hip.so that lets the
 * CLI's host threads (StretcherProcessor twin, WindowQueue, the hot-swap watcher, AudioBus drain) run under
 * ThreadSanitizer on a machine without a GPU (rocoder_amd/csrc/host/sanitize.mk, tools/run_sanitizers.sh).
 * It computes NOTHING: a "window" is the next window_len input samples, zero-padded at the end of the stream. Only the
 * entry points the CLI calls exist. */
#include <stdlib.h>
#include <string.h>

#include "rocoder_hip.h"

struct chan {
    float *x;
    size_t len, cap, pos;
    int closed;
};
struct rc_engine {
    rc_config cfg;
    struct chan *ch;
};
const char *rc_last_error(void) { return "stub engine"; }
int rc_engine_create(const rc_config *cfg, rc_engine **out) {
    if (!cfg || !out || cfg->window_len < 2 || cfg->channels == 0) return RC_EINVAL;
    rc_engine *e = (rc_engine *)calloc(1, sizeof *e);
    e->cfg = *cfg;
    e->ch = (struct chan *)calloc(cfg->channels, sizeof(struct chan));
    *out = e;
    return RC_OK;
}
void rc_engine_destroy(rc_engine *e) {
    if (!e) return;
    for (uint32_t c = 0; c < e->cfg.channels; ++c) free(e->ch[c].x);
    free(e->ch);
    free(e);
}
int rc_engine_get_params(const rc_engine *e, rc_params *p) {
    memset(p, 0, sizeof *p);
    p->window_len = e->cfg.window_len;
    p->half_window_len = e->cfg.window_len / 2;
    p->window_out_len = e->cfg.window_len;
    p->hops_per_window = 2;
    p->sample_step_len = e->cfg.window_len / 2;
    p->samples_needed_per_window = e->cfg.window_len;
    return RC_OK;
}
int rc_engine_push_input(rc_engine *e, uint32_t c, const float *s, size_t n) {
    struct chan *k = &e->ch[c];
    if (k->len + n > k->cap) {
        k->cap = (k->len + n) * 2;
        k->x = (float *)realloc(k->x, k->cap * sizeof(float));
    }
    memcpy(k->x + k->len, s, n * sizeof(float));
    k->len += n;
    return RC_OK;
}
int rc_engine_close_input(rc_engine *e, uint32_t c) {
    e->ch[c].closed = 1;
    return RC_OK;
}
int rc_engine_is_done(const rc_engine *e, uint32_t c) { return e->ch[c].closed && e->ch[c].pos >= e->ch[c].len; }
size_t rc_engine_channel_bound(const rc_engine *e) { (void)e; return 1; }
int rc_engine_next_window(rc_engine *e, uint32_t c, float *out, size_t cap, size_t *n_out) {
    struct chan *k = &e->ch[c];
    const size_t N = e->cfg.window_len;
    if (cap < N) return RC_ECAPACITY;
    if (!k->closed && k->len - k->pos < N) return RC_WOULD_BLOCK;
    const size_t have = k->len - k->pos < N ? k->len - k->pos : N;
    memcpy(out, k->x + k->pos, have * sizeof(float));
    memset(out + have, 0, (N - have) * sizeof(float));
    k->pos += N;
    *n_out = N;
    return RC_OK;
}

/* --devices (rc_multi_*): not part of what the TSan run exercises - the CLI only needs the symbols to link */
size_t rc_offline_output_len(const rc_config *cfg, size_t in_len) { (void)cfg; return in_len; }
int rc_multi_create(const rc_config *cfg, const int32_t *device_ids, uint32_t n_devices, rc_multi **out) {
    (void)cfg; (void)device_ids; (void)n_devices; (void)out;
    return RC_EUNSUPPORTED;
}
void rc_multi_destroy(rc_multi *m) { (void)m; }
int rc_multi_stretch_host(rc_multi *m, const float *const *in, size_t in_len, float *const *out, size_t out_cap,
                          size_t *out_len) {
    (void)m; (void)in; (void)in_len; (void)out; (void)out_cap; (void)out_len;
    return RC_EUNSUPPORTED;
}
