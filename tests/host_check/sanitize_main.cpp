// sanitize_main.cpp -- AddressSanitizer / UndefinedBehaviorSanitizer run (CPU build only) over the product's lane-local device
// functions (csrc/ccsp_rules.h, through host_check.cpp) and over the C oracle: random positions through move generation (all
// three host formulations), step, planes, progress, the evaluator tables, the random stream and the samplers; whole oracle games.
// TEST BUILD ONLY.  Built and run by tests/test_device_logic_on_host.py::test_sanitizers_find_nothing.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "host_check.cpp"

extern "C" {
int orc_movegen(const uint8_t *pos12, int player, uint8_t *moves);
void orc_randomised_pos12(uint64_t seed, uint64_t game, uint8_t *pos12);
void orc_initial_pos12(uint8_t *pos12);
int orc_random_move(const uint8_t *pos12, int player, uint64_t seed, uint64_t game, uint32_t ply, int *id, int *dest);
int orc_step(const uint8_t *pos12, const uint8_t *last4, int player, int id, int dest, uint8_t *npos12, uint8_t *nlast4, uint8_t *nboard);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2000;
    long moves_total = 0, mismatches = 0;
    for (int g = 0; g < n; g++) {
        uint8_t pos[12], last[4] = {255, 255, 255, 255};
        if (g % 3 == 0) orc_randomised_pos12(7, (uint64_t)g, pos); else orc_initial_pos12(pos);
        int player = 1;
        for (int ply = 0; ply < 24; ply++) {
            uint8_t a[126][2], b[126][2], c[126][2], d[126][2];
            uint64_t masks[6];
            const int ka = hc_movegen(pos, player, &a[0][0], masks);
            const int kb = hc_movegen_lines(pos, player, &b[0][0]);
            const int kc = hc_movegen_stack(pos, player, &c[0][0]);
            const int kd = orc_movegen(pos, player, &d[0][0]);
            if (ka != kb || ka != kc || ka != kd || memcmp(a, b, 2 * ka) || memcmp(a, c, 2 * ka) || memcmp(a, d, 2 * ka)) mismatches++;
            moves_total += ka;
            (void)hc_stack_depth(pos, player);
            (void)hc_progress(pos, player);
            uint8_t planes[343], planes2[343];
            hc_planes(pos, last, player, planes);
            hc_planes_scatter(pos, last, player, planes2);
            if (memcmp(planes, planes2, 343)) mismatches++;
            double p[294]; float v;
            hc_hash_eval(pos, player, p, &v);
            hc_forward_eval(pos, player, p, &v);
            if (ka == 0) break;
            int id, dest;
            (void)orc_random_move(pos, player, 11, (uint64_t)g, (uint32_t)ply, &id, &dest);
            uint8_t np[12], nl[4], np2[12], nl2[4], brd[147];
            const int w1 = hc_step(pos, last, player, id, dest, np, nl);
            const int w2 = orc_step(pos, last, player, id, dest, np2, nl2, brd);
            if (w1 != w2 || memcmp(np, np2, 12) || memcmp(nl, nl2, 4)) mismatches++;
            memcpy(pos, np, 12); memcpy(last, nl, 4);
            if (w1) break;
            player = 3 - player;
        }
        (void)hc_gamma(3, (uint64_t)g, 5, (uint32_t)(g % 90), 0.03);
        (void)hc_choice(hc_rng(1, (uint64_t)g, 2, 3, 4, 5), (uint32_t)(g % 126 + 1));
        (void)hc_det_exp(hc_det_log(1.0 + g * 0.37));
    }
    (void)hc_div_sweep(64, 2);
    printf("positions %d  moves %ld  mismatches %ld\n", n, moves_total, mismatches);
    return mismatches ? 1 : 0;
}
