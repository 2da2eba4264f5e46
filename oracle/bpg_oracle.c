/*
 * bpg_oracle.c -- CPU ORACLE (test infrastructure, see bpg_oracle.h).
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  Deliberately written the way the reference computes:
 * float64 matrix, nested loops.  Do not "optimise" this file -- it is the
 * checker, and bench.py times it only as the labelled cpu_baseline "port".
 */
#include "bpg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* gym_blocks/envs/fetch_env.py:19-28 (evaluated in Python float64 order) */
static double table_w(void) { double w = 0.25; w -= 0.05 / 2; return w; }
static double table_h(void) { double h = 0.35; h -= 0.05 / 2; return h; }
static double table_x(void) { return 1.05 + 0.25; }
static double table_y(void) { return 0.40 + 0.35; }

/* fetch_env.py:30-32 */
int orc_out_of_table(double x, double y)
{
    return fabs(x - table_x()) > table_w() || fabs(y - table_y()) > table_h();
}

/* fetch_env.py:106-117 */
int orc_geom2objid(const char *name)
{
    if (name != NULL) {
        if (strstr(name, "finger") != NULL)
            return 0;
        else if (strcmp(name, "table") == 0)
            return 1;
        else if (strstr(name, "object") != NULL)
            return atoi(name + 6) + 2;      /* int(name[6:]) + 2 */
    }
    return -1;
}

/* fetch_env.py:323-326 (GripperTouch), :360-363 (BlocksTouch), :434-441
 * (Choose; shuffle commented out :438), :632-639 (Variation, max_num_blocks=4
 * :523), :772-775 (ToppleTower). */
int orc_sample_colors(int variant, int *c)
{
    switch (variant) {
    case ORC_GRIPPER_TOUCH:
        c[0] = ORC_BLUE; c[1] = ORC_GREY; c[2] = ORC_GREEN;
        return 3;
    case ORC_BLOCKS_TOUCH:
    case ORC_BLOCKS_TOUCH_CURRICULUM:
        c[0] = ORC_GREY; c[1] = ORC_GREY; c[2] = ORC_GREEN; c[3] = ORC_BLUE;
        return 4;
    case ORC_BLOCKS_TOUCH_CHOOSE:
    case ORC_BLOCKS_TOUCH_CHOOSE_CURRICULUM:
        c[0] = ORC_GREY; c[1] = ORC_GREY; c[2] = ORC_GREEN; c[3] = ORC_BLUE; c[4] = ORC_GREY;
        return 5;
    case ORC_BLOCKS_TOUCH_VARIATION:
        c[0] = ORC_GREY; c[1] = ORC_GREY; c[2] = ORC_GREEN; c[3] = ORC_BLUE;
        c[4] = ORC_GREY; c[5] = ORC_GREY;
        return 6;
    case ORC_TOPPLE_TOWER:
        c[0] = ORC_RED; c[1] = ORC_GREEN; c[2] = ORC_GREY; c[3] = ORC_GREY;
        c[4] = ORC_GREY; c[5] = ORC_BLUE;
        return 6;
    }
    return 0;
}

/* fetch_env.py:260-273; Variation :682-695 always uses side 6 */
void orc_sample_goal(const int *C, int side, long long *goal)
{
    int k = 0;
    for (int i = 0; i < side; i++) {
        for (int j = 0; j < side; j++) {
            if ((C[i] == ORC_RED && C[j] == ORC_BLUE) || (C[i] == ORC_BLUE && C[j] == ORC_RED))
                goal[k++] = -1;
            else if ((C[i] == ORC_GREEN && C[j] == ORC_BLUE) || (C[i] == ORC_BLUE && C[j] == ORC_GREEN))
                goal[k++] = 1;
            else
                goal[k++] = 0;
        }
    }
}

/* fetch_env.py:64-86 (constructor state) + robot_env.py:36 (goal sampled once
 * in __init__); Variation keeps max_num_blocks + 2 = 6 (:523-527). */
void orc_env_init(orc_env *e, int variant, int max_episode_steps)
{
    memset(e, 0, sizeof(*e));
    e->variant = variant;
    e->is_variation = (variant == ORC_BLOCKS_TOUCH_VARIATION);
    e->num_objs = orc_sample_colors(variant, e->obj_colors);
    e->side = e->num_objs;
    for (int k = 0; k < ORC_MAX_G; k++)
        e->achieved_goal[k] = -1.0;            /* :78 */
    e->has_succeeded = 0;                      /* :80 */
    orc_sample_goal(e->obj_colors, e->side, e->goal);
    e->elapsed_steps = 0;
    e->max_episode_steps = max_episode_steps;
}

/* robot_env.py:71-82 -> _reset_sim fetch_env.py:247-255: colours re-sampled
 * (constant), has_succeeded cleared, achieved_goal untouched (Q1).
 * Variation _reset_sim :646-679: num_grey = np_random.randint(3) (explicit
 * input here), num_objs = 4 + num_grey, matrix re-created as -1 (6x6).
 * Then goal = _sample_goal() (robot_env.py:80).  TimeLimit.reset zeroes the
 * step counter (third-party). */
void orc_env_reset(orc_env *e, int num_grey)
{
    if (e->is_variation) {
        e->num_objs = 4 + num_grey;
        e->side = 6;
        for (int k = 0; k < ORC_MAX_G; k++)
            e->achieved_goal[k] = -1.0;
    }
    orc_sample_colors(e->variant, e->obj_colors);
    e->has_succeeded = 0;
    orc_sample_goal(e->obj_colors, e->side, e->goal);
    e->elapsed_steps = 0;
}

/* fetch_env.py:154-157: entries == 1 -> 0 for i,j < num_objs (row stride is
 * the allocated side); :158-167: contacts with both geoms mapped set 1. */
static void demote(orc_env *e)
{
    for (int i = 0; i < e->num_objs; i++)
        for (int j = 0; j < e->num_objs; j++)
            if (e->achieved_goal[i * e->side + j] == 1)
                e->achieved_goal[i * e->side + j] = 0;
}

void orc_step_callback(orc_env *e, const int *id2obj, const int *contacts, int ncon)
{
    demote(e);
    for (int i = 0; i < ncon; i++) {
        int obj1 = id2obj[contacts[2 * i]];
        int obj2 = id2obj[contacts[2 * i + 1]];
        if (obj1 != -1 && obj2 != -1) {
            e->achieved_goal[obj1 * e->side + obj2] = 1;
            e->achieved_goal[obj2 * e->side + obj1] = 1;
        }
    }
}

int orc_pair_index(int i, int j)
{
    if (i > j) { int t = i; i = j; j = t; }
    return i * ORC_MAX_OBJS - i * (i - 1) / 2 + (j - i);
}

uint32_t orc_contacts_to_mask(const int *id2obj, const int *contacts, int ncon)
{
    uint32_t m = 0;
    for (int i = 0; i < ncon; i++) {
        int a = id2obj[contacts[2 * i]], b = id2obj[contacts[2 * i + 1]];
        if (a != -1 && b != -1)
            m |= 1u << orc_pair_index(a, b);
    }
    return m;
}

void orc_step_callback_mask(orc_env *e, uint32_t mask)
{
    demote(e);
    for (int i = 0; i < e->side; i++)
        for (int j = i; j < e->side; j++)
            if (mask & (1u << orc_pair_index(i, j))) {
                e->achieved_goal[i * e->side + j] = 1;
                e->achieved_goal[j * e->side + i] = 1;
            }
}

/* fetch_env.py:119-124 */
int orc_check_goal(const orc_env *e)
{
    for (int i = 0; i < e->num_objs; i++)
        for (int j = 0; j < e->num_objs; j++)
            if (e->achieved_goal[i * e->side + j] != e->achieved_goal[j * e->side + i])
                return 0;
    return 1;
}

/* np.sum over the last axis of a float array = NumPy's pairwise add-reduction (third-party NumPy,
 * numpy/_core/src/umath/loops_utils.h.src, restated; tests/test_oracle_golden.py checks this restatement against
 * NumPy itself and against tests/golden/compute_reward_order.npz): below 8 elements a plain loop; up to 128
 * elements eight strided partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus the n % 8 tail; above
 * that a split at n/2 rounded down to a multiple of 8.  The order only matters for rows that are not ternary. */
static double orc_pairwise_f64(const double *a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        int i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return orc_pairwise_f64(a, n2) + orc_pairwise_f64(a + n2, n - n2);
}

static float orc_pairwise_f32(const float *a, int n)
{
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        int i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return orc_pairwise_f32(a, n2) + orc_pairwise_f32(a + n2, n - n2);
}

/* fetch_env.py:141-143: d = np.sum(ag * g, axis=-1); c = np.count_nonzero(g, axis=-1);
 * -(d != c).astype(float32): -1.0f on mismatch, -0.0f on match (quirk Q2).  float64 arrays. */
float orc_compute_reward(const double *ag, const double *g, int len)
{
    double prod[1024];
    long long c = 0;
    if (len > 1024) len = 1024;
    for (int k = 0; k < len; k++) {
        prod[k] = ag[k] * g[k];
        if (g[k] != 0.0)
            c++;
    }
    const double d = orc_pairwise_f64(prod, len);
    float ne = (d != (double)c) ? 1.0f : 0.0f;
    return -ne;
}

void orc_compute_reward_batch(const double *ag, const double *g, int n, int len, float *out)
{
    for (int r = 0; r < n; r++)
        out[r] = orc_compute_reward(ag + (size_t)r * len, g + (size_t)r * len, len);
}

/* the same on float32 arrays (HER's replay buffers, config.py:110-111): float32 products, float32 pairwise sum;
 * d (float32) != c (int64) compares in float64 */
void orc_compute_reward_f32_batch(const float *ag, const float *g, int n, int len, float *out, float *dsum)
{
    float prod[1024];
    if (len > 1024) len = 1024;
    for (int r = 0; r < n; r++) {
        long long c = 0;
        for (int k = 0; k < len; k++) {
            prod[k] = ag[(size_t)r * len + k] * g[(size_t)r * len + k];
            if (g[(size_t)r * len + k] != 0.0f)
                c++;
        }
        const float d = orc_pairwise_f32(prod, len);
        if (dsum) dsum[r] = d;
        out[r] = -(((double)d != (double)c) ? 1.0f : 0.0f);
    }
}

/* the sums themselves (for pinning the order against NumPy) */
void orc_rowsum_f64(const double *a, int n, int len, double *out)
{
    for (int r = 0; r < n; r++) out[r] = orc_pairwise_f64(a + (size_t)r * len, len);
}

void orc_rowsum_f32(const float *a, int n, int len, float *out)
{
    for (int r = 0; r < n; r++) out[r] = orc_pairwise_f32(a + (size_t)r * len, len);
}

/* fetch_env.py:220 / Variation :610-613 (padding to 36 is empty: side is 6) */
void orc_get_achieved_goal(const orc_env *e, double *out)
{
    for (int k = 0; k < e->side * e->side; k++)
        out[k] = e->achieved_goal[k];
}

/* robot_env.py:58: np.clip(action, -1, 1) (float32 Box bounds); fetch_env.py:170-185:
 * pos_ctrl *= 0.05 in place on the float32 array, fixed quaternion, both finger targets =
 * action[3] (zeros if block_gripper), concatenated into a float64 vector. */
void orc_set_action(const float *action, int block_gripper, double *out9)
{
    float a[4];
    for (int k = 0; k < 4; k++)
        a[k] = action[k] < -1.0f ? -1.0f : (action[k] > 1.0f ? 1.0f : action[k]);
    for (int k = 0; k < 3; k++) {
        float p = a[k] * 0.05f;              /* float32 array *= python float -> float32 product */
        out9[k] = (double)p;
    }
    out9[3] = 1.; out9[4] = 0.; out9[5] = 1.; out9[6] = 0.;
    out9[7] = out9[8] = block_gripper ? 0.0 : (double)a[3];
}

/* fetch_env.py:187-228: obs = [grip_pos, gripper_state, grip_velp*dt, gripper_vel*dt] +
 * per block [pos, pos - grip_pos, euler, velp*dt - grip_velp*dt, velr*dt].
 * Variation :567-621: leading num_blocks, per block + one_hot_color (:34-37), zero
 * padding up to max_num_blocks = 4 blocks of 19 features. */
int orc_get_obs(const orc_env *e, const orc_phys *p, double dt, double *obs)
{
    int n = 0;
    const int num_blocks = e->num_objs - 2;
    double grip_velp[3], gripper_vel[2];
    for (int c = 0; c < 3; c++) grip_velp[c] = p->grip_velp[c] * dt;        /* :191 */
    for (int c = 0; c < 2; c++) gripper_vel[c] = p->gripper_vel[c] * dt;    /* :197 */
    if (e->is_variation) obs[n++] = (double)num_blocks;                     /* :580 */
    for (int c = 0; c < 3; c++) obs[n++] = p->grip_pos[c];
    for (int c = 0; c < 2; c++) obs[n++] = p->gripper_state[c];
    for (int c = 0; c < 3; c++) obs[n++] = grip_velp[c];
    for (int c = 0; c < 2; c++) obs[n++] = gripper_vel[c];
    for (int i = 0; i < num_blocks; i++) {
        const double *pos = p->obj_pos + 3 * i, *rot = p->obj_rot + 3 * i;
        double velp[3], velr[3];
        for (int c = 0; c < 3; c++) {
            velp[c] = p->obj_velp[3 * i + c] * dt;                           /* :207 */
            velr[c] = p->obj_velr[3 * i + c] * dt;                           /* :208 */
        }
        for (int c = 0; c < 3; c++) velp[c] -= grip_velp[c];                 /* :211 */
        for (int c = 0; c < 3; c++) obs[n++] = pos[c];
        for (int c = 0; c < 3; c++) obs[n++] = pos[c] - p->grip_pos[c];      /* :210 */
        for (int c = 0; c < 3; c++) obs[n++] = rot[c];
        for (int c = 0; c < 3; c++) obs[n++] = velp[c];
        for (int c = 0; c < 3; c++) obs[n++] = velr[c];
        if (e->is_variation)
            for (int c = 0; c < ORC_NUM_COLORS; c++)
                obs[n++] = (e->obj_colors[i + 2] == c) ? 1.0 : 0.0;          /* :599 */
    }
    if (e->is_variation)
        for (int k = 0; k < 19 * (4 - num_blocks); k++) obs[n++] = 0.0;      /* :606-607 */
    return n;
}

static float reward_of_state(const orc_env *e)
{
    double g[ORC_MAX_G];
    int len = e->side * e->side;
    for (int k = 0; k < len; k++)
        g[k] = (double)e->goal[k];
    return orc_compute_reward(e->achieved_goal, g, len);
}

/* fetch_env.py:275-281: ignores its arguments, r == 0 (true for -0.0) latches */
int orc_is_success(orc_env *e)
{
    float r = reward_of_state(e);
    if (r == 0)
        e->has_succeeded = 1;
    return e->has_succeeded;
}

/* robot_env.py:57-69 after sim.step(): _step_callback, _get_obs (asserts
 * _check_goal fetch_env.py:219), info['is_success'], reward; env done=False.
 * gym TimeLimit (max_episode_steps=50, gym_blocks/__init__.py:10) ORs in
 * elapsed_steps >= max (third-party behaviour). */
static void finish_step(orc_env *e, float *reward, int *done, int *is_success)
{
    if (!orc_check_goal(e))
        abort();
    *is_success = orc_is_success(e);
    *reward = reward_of_state(e);
    e->elapsed_steps += 1;
    *done = (e->elapsed_steps >= e->max_episode_steps) ? 1 : 0;
}

void orc_env_step(orc_env *e, const int *id2obj, const int *contacts, int ncon,
                  float *reward, int *done, int *is_success)
{
    orc_step_callback(e, id2obj, contacts, ncon);
    finish_step(e, reward, done, is_success);
}

void orc_env_step_mask(orc_env *e, uint32_t mask, float *reward, int *done, int *is_success)
{
    orc_step_callback_mask(e, mask);
    finish_step(e, reward, done, is_success);
}

/* policy_gradient/rollout.py:255-258 -- at step t the reference appends r_t and
 * then, for every earlier t_, does returns[t_] += gamma ** (t - t_) * r_t.
 * So G[t_] accumulates r_t_ first, then the later terms in increasing t. */
void orc_discounted_return(const double *r, int T, double gamma, double *G)
{
    for (int t = 0; t < T; t++) {
        G[t] = r[t];
        for (int t_ = 0; t_ < t; t_++)
            G[t_] += pow(gamma, (double)(t - t_)) * r[t];
    }
}

/* ---- reset placement, fetch_env.py:328-336, 370-399, 448-517, 697-764, 777-787 ---- */
typedef struct { const orc_placement *p; int ie, iz, iu, fail; } draws;

static double next_eu(draws *d) { if (d->ie >= d->p->n_eu) { d->fail = 1; return 0.5; } return d->p->eu[d->ie++]; }
static double next_gz(draws *d) { if (d->iz >= d->p->n_gz) { d->fail = 1; return 1.0; } return d->p->gz[d->iz++]; }
static double next_gu(draws *d) { if (d->iu >= d->p->n_gu) { d->fail = 1; return 0.5; } return d->p->gu[d->iu++]; }
/* RandomState.uniform(low, high): low + (high - low) * random_sample() */
static double unif(double lo, double hi, double u) { return lo + (hi - lo) * u; }
/* np.linalg.norm([x, y]) = sqrt(x.dot(x)); ddot of 2 elements as the recording container computes it */
static double norm2(double x, double y) { return sqrt(fma(y, y, x * x)); }

/* direction = normal(size=2); direction /= norm; mag = uniform(lo, hi); c + direction*mag
 * (fetch_env.py:390-393 and the same four lines at :490-493, :507-510, :734-737) */
static void polar_about(draws *d, double cx, double cy, double lo, double hi, double *x, double *y)
{
    double dx = next_gz(d), dy = next_gz(d);
    double n = norm2(dx, dy);
    dx = dx / n; dy = dy / n;
    double mag = unif(lo, hi, next_gu(d));
    *x = cx + dx * mag;
    *y = cy + dy * mag;
}

int orc_randomize_objects(const orc_env *e, const orc_placement *p, double *xy, int *used)
{
    draws d = {p, 0, 0, 0, 0};
    const double MIN_BLOCK_DIST = 1.5 * 0.05;                       /* fetch_env.py:19-20 */
    const double gx = p->grip_x, gy = p->grip_y;
    double x, y;
    switch (e->variant) {
    case ORC_GRIPPER_TOUCH:                                         /* :328-336 */
    case ORC_TOPPLE_TOWER: {                                        /* :777-787 */
        x = gx; y = gy;
        while (!d.fail && norm2(x - gx, y - gy) < 0.1) {
            x = gx + unif(-p->obj_range, p->obj_range, next_eu(&d));
            y = gy + unif(-p->obj_range, p->obj_range, next_eu(&d));
        }
        int nb = (e->variant == ORC_TOPPLE_TOWER) ? 4 : 1;
        for (int i = 0; i < nb; i++) { xy[2 * i] = x; xy[2 * i + 1] = y; }
        break;
    }
    case ORC_BLOCKS_TOUCH:
    case ORC_BLOCKS_TOUCH_CURRICULUM: {                             /* :370-399 */
        double r = p->test ? p->max_obj_range : p->obj_range;
        double x0 = gx + unif(-r / 2, r / 2, next_eu(&d));
        double y0 = gy + unif(-r / 2, r / 2, next_eu(&d));
        do {
            polar_about(&d, x0, y0, MIN_BLOCK_DIST, r, &x, &y);
        } while (!d.fail && orc_out_of_table(x, y));
        xy[0] = x0; xy[1] = y0; xy[2] = x; xy[3] = y;
        break;
    }
    case ORC_BLOCKS_TOUCH_CHOOSE:
    case ORC_BLOCKS_TOUCH_CHOOSE_CURRICULUM: {                      /* :448-517 */
        double r, wrong_r, max_wrong, min_r;
        if (p->test || p->challenge) { r = p->max_obj_range; wrong_r = 0; }
        else { r = p->obj_range; wrong_r = p->wrong_obj_range; }
        if (p->challenge) { max_wrong = 0.04; min_r = 0.15; }
        else { min_r = MIN_BLOCK_DIST; max_wrong = p->max_obj_range; }
        const int green = 0, blue = 1, wrong = 2;                   /* colours [GREEN, BLUE, GREY] :437 */
        double bx, by, gx2, gy2;
        do {
            bx = gx + unif(-r / 2, r / 2, next_eu(&d));
            by = gy + unif(-r / 2, r / 2, next_eu(&d));
        } while (!d.fail && orc_out_of_table(bx, by));
        do {
            polar_about(&d, bx, by, min_r, r, &gx2, &gy2);
        } while (!d.fail && orc_out_of_table(gx2, gy2));
        double cx = (bx + gx2) / 2.0, cy = (by + gy2) / 2.0;
        do {
            polar_about(&d, cx, cy, wrong_r, max_wrong, &x, &y);
        } while (!d.fail && (orc_out_of_table(x, y) || norm2(x - bx, y - by) < MIN_BLOCK_DIST ||
                             norm2(x - gx2, y - gy2) < MIN_BLOCK_DIST));
        xy[2 * blue] = bx; xy[2 * blue + 1] = by;
        xy[2 * green] = gx2; xy[2 * green + 1] = gy2;
        xy[2 * wrong] = x; xy[2 * wrong + 1] = y;
        break;
    }
    case ORC_BLOCKS_TOUCH_VARIATION: {                              /* :697-764 */
        const int num_blocks = e->num_objs - 2;
        double r = p->test ? p->max_obj_range : p->obj_range;
        const int green = 0, blue = 1;                              /* colours [GREEN, BLUE, GREY, GREY] :635 */
        double px[4], py[4];
        int np_ = 0;
        double bx, by, gx2, gy2;
        do {
            bx = gx + unif(-r / 2, r / 2, next_eu(&d));
            by = gy + unif(-r / 2, r / 2, next_eu(&d));
        } while (!d.fail && orc_out_of_table(bx, by));
        do {
            polar_about(&d, bx, by, MIN_BLOCK_DIST, r, &gx2, &gy2);
        } while (!d.fail && orc_out_of_table(gx2, gy2));
        xy[2 * blue] = bx; xy[2 * blue + 1] = by;
        xy[2 * green] = gx2; xy[2 * green + 1] = gy2;
        px[np_] = bx; py[np_++] = by;
        px[np_] = gx2; py[np_++] = gy2;
        for (int i = 0; i < num_blocks; i++) {
            if (i == blue || i == green) continue;
            int again;
            do {
                again = 0;
                x = table_x() + unif(-table_w(), table_w(), next_eu(&d));   /* _sample_from_table :88-90 */
                y = table_y() + unif(-table_h(), table_h(), next_eu(&d));
                int hit = 0;
                for (int k = 0; k < np_; k++)
                    if (norm2(x - px[k], y - py[k]) < MIN_BLOCK_DIST) { hit = 1; break; }
                again = hit ? 1 : orc_out_of_table(x, y);           /* for-else :753-758 */
            } while (!d.fail && again);
            xy[2 * i] = x; xy[2 * i + 1] = y;
            px[np_] = x; py[np_++] = y;
        }
        break;
    }
    }
    if (used) { used[0] = d.ie; used[1] = d.iz; used[2] = d.iu; }
    return d.fail;
}

/* ---- build-defined synthetic contact source (no reference counterpart) ---- */
uint32_t orc_hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

static uint32_t orc_mulfold(uint32_t x, uint32_t k)
{
    uint64_t p = (uint64_t)x * (uint64_t)k;
    return (uint32_t)(p >> 32) ^ (uint32_t)p;
}

static uint32_t orc_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* Per-env key of the build-defined generator.  The reference seeds env idx with seed + 1000 * idx
 * (RolloutStudent.seed, gym_blocks/rollout.py:206-210): that arithmetic is mirrored on the global env index
 * (64-bit, wrapping); gym's seeding behind it (SHA-512 into MT19937) is third-party and replaced by
 *   key(es, salt) = h((lo32(es) ^ salt) ^ h(hi32(es) + 0x9E3779B9)),   es = seed + 1000 * env   (h = lowbias32) */
uint32_t orc_env_key(uint64_t seed, uint64_t env_id, uint32_t salt)
{
    const uint64_t es = seed + 1000ull * env_id;
    return orc_hash32(((uint32_t)es ^ salt) ^ orc_hash32((uint32_t)(es >> 32) + 0x9E3779B9U));
}

/* np_random.randint(3) stand-in of the Variation reset (fetch_env.py:647): generator-drawn reset number
 * `counter` of the handle */
int orc_num_grey_draw(uint64_t seed, uint64_t env_id, uint32_t counter)
{
    return (int)(orc_hash32(orc_env_key(seed, env_id, 0xA511E9B3U) + counter) % 3u);
}

/* The spec of the source, stated here independently of the kernels:
 *   ka = key(seed + 1000 * env, 0),  kb = h(ka ^ 0x85EBCA6B)
 *   u  = ka + (t + 1) * 0x9E3779B9
 *   a  = fold((u      ^ (u      >> 15)) * 0xD6E8FEB9)     fold(p) = hi32(p) ^ lo32(p) of the
 *   b  = fold(((u^kb) ^ ((u^kb) >> 15)) * 0xCA5A8263)     64-bit product
 *   pair j touches iff U_j < 5 for the bit-sliced 5-bit uniform U_j whose "digit is zero" bits are
 *   A_j = a[j], B_j = b[j], C_j = (a >> 21 | (b >> 21) << 11)[j], D_j = rotl(a,7)[j] & rotl(b,13)[j]:
 *   A & B & (C | D)  ->  exactly 5/32 per pair.                                              */
uint32_t orc_synth_mask(uint64_t seed, uint64_t env_id, uint32_t t, int num_objs)
{
    const uint32_t M21 = (1u << 21) - 1;
    uint32_t ka = orc_env_key(seed, env_id, 0u);
    uint32_t kb = orc_hash32(ka ^ 0x85EBCA6BU);
    uint32_t u = ka + (t + 1u) * 0x9E3779B9U;
    uint32_t x = u ^ (u >> 15);
    uint32_t y = (u ^ kb) ^ ((u ^ kb) >> 15);
    uint32_t a = orc_mulfold(x, 0xD6E8FEB9U), b = orc_mulfold(y, 0xCA5A8263U);
    uint32_t acc = 0;
    for (int j = 0; j < 21; j++) {
        int A = (a >> j) & 1, B = (b >> j) & 1;
        int Cj = j < 11 ? (a >> (21 + j)) & 1 : (b >> (21 + (j - 11))) & 1;
        int Dj = ((orc_rotl(a, 7) >> j) & 1) & ((orc_rotl(b, 13) >> j) & 1);
        if (A && B && (Cj || Dj)) acc |= 1u << j;
    }
    acc &= M21;
    uint32_t valid = 0, forced = 0;
    for (int i = 0; i < num_objs; i++)
        for (int j = i + 1; j < num_objs; j++)
            valid |= 1u << orc_pair_index(i, j);
    for (int k = 2; k < num_objs; k++)
        forced |= 1u << orc_pair_index(1, k);
    return (acc & valid) | forced;
}

/* rollout.py:60-64 reset_all_rollouts, :75-172 generate_rollouts.  The policy
 * and the physics are outside the boundary: the step's contacts arrive as
 * masks[t*B+b].  float64 -> float32 stores as rollout.py:82-83,117-118,149-150.
 * Episode arrays batch-major as convert_episode_to_batch_major (util.py:118-128). */
double orc_generate_rollouts(orc_env *envs, int B, int T, const uint32_t *masks,
                             const int *num_grey,
                             float *ag, float *g, float *info_is_success, float *reward_tb)
{
    double succ_sum = 0.0;
    for (int b = 0; b < B; b++) {
        orc_env *e = &envs[b];
        orc_env_reset(e, num_grey ? num_grey[b] : 0);
        int dimg = e->side * e->side;
        double cur[ORC_MAX_G];
        orc_get_achieved_goal(e, cur);                        /* initial_ag */
        int last_success = 0;
        for (int t = 0; t < T; t++) {
            float r; int done, succ;
            for (int k = 0; k < dimg; k++) {
                ag[((size_t)b * (T + 1) + t) * dimg + k] = (float)cur[k];
                g[((size_t)b * T + t) * dimg + k] = (float)e->goal[k];
            }
            orc_env_step_mask(e, masks[(size_t)t * B + b], &r, &done, &succ);
            orc_get_achieved_goal(e, cur);
            info_is_success[(size_t)b * T + t] = (float)succ;
            if (reward_tb)
                reward_tb[(size_t)t * B + b] = r;
            last_success = succ;
        }
        for (int k = 0; k < dimg; k++)
            ag[((size_t)b * (T + 1) + T) * dimg + k] = (float)cur[k];
        succ_sum += last_success;                            /* rollout.py:163-167 */
    }
    return succ_sum / B;
}

uint64_t orc_bench_rollout(int variant, int B, int T, uint64_t seed)
{
    uint64_t sum = 0;
    orc_env e;
    for (int b = 0; b < B; b++) {
        orc_env_init(&e, variant, T);
        orc_env_reset(&e, 2);
        for (int t = 0; t < T; t++) {
            float r; int done, succ;
            uint32_t m = orc_synth_mask(seed, (uint64_t)b, (uint32_t)t, e.num_objs);
            orc_env_step_mask(&e, m, &r, &done, &succ);
            sum += (uint64_t)(r == 0) + 2u * (uint64_t)succ + 4u * (uint64_t)done;
        }
    }
    return sum;
}

void orc_fill_masks(uint64_t seed, uint64_t env_base, int B, int T, int num_objs, uint32_t *masks)
{
    for (int t = 0; t < T; t++)
        for (int b = 0; b < B; b++)
            masks[(size_t)t * B + b] = orc_synth_mask(seed, env_base + (uint64_t)b, (uint32_t)t, num_objs);
}

uint64_t orc_bench_rollout_masks(int variant, int B, int T, int num_grey, const uint32_t *masks,
                                 float *reward_tb, uint8_t *flags_tb)
{
    uint64_t sum = 0;
    orc_env e;
    for (int b = 0; b < B; b++) {
        orc_env_init(&e, variant, T);
        orc_env_reset(&e, num_grey);
        for (int t = 0; t < T; t++) {
            float r; int done, succ;
            orc_env_step_mask(&e, masks[(size_t)t * B + b], &r, &done, &succ);
            reward_tb[(size_t)t * B + b] = r;
            flags_tb[(size_t)t * B + b] = (uint8_t)(succ | (done << 1) | ((r == 0) << 2));
        }
        sum += (uint64_t)e.has_succeeded;
    }
    return sum;
}
