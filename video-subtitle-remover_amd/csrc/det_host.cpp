// Host side of the detector's post-process: border following over the thresholded probability map -- what cv2.findContours does
// for PaddleX's DBPostProcess (reference call chain: backend/tools/subtitle_detect.py:41-63 -> paddleocr TextDetection.predict ->
// DBPostProcess.boxes_from_bitmap -> cv2.findContours(RETR_LIST, CHAIN_APPROX_SIMPLE); Suzuki & Abe 1985, 8-connected foreground).
// The GPU runs the post-process of maps without holes (det_kernels.hip); a map with a hole, with more components than the device
// record list or with a very tall component is finished on the host (backend/tools/ocr_det.db_postprocess), and this is its inner
// loop: the Python statement of the same walk took 350 ms on a noise map, this takes a millisecond.  Host code by nature -- the
// reference's own post-process is cv2 / pyclipper on the CPU -- and no substitute for a kernel: nothing here touches a tensor the
// GPU computes on.
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "../../include/vsr_hip.h"

extern "C" int vsr_host_trace_borders(const uint8_t* bitmap, int H, int W, int32_t* points_xy, int64_t cap_points, int64_t* border_start,
                                      int32_t cap_borders, int32_t* n_borders, int64_t* n_points)
{
    if (!bitmap || !points_xy || !border_start || !n_borders || !n_points || H <= 0 || W <= 0 || cap_points <= 0 || cap_borders <= 0)
        return VSR_ERR_ARG;
    const int Wp = W + 2;
    std::vector<int32_t> g((size_t)(H + 2) * Wp, 0);          // zero frame: the image border is background, as in cv2
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) g[(size_t)(y + 1) * Wp + x + 1] = bitmap[(size_t)y * W + x] ? 1 : 0;
    // the 8 neighbours counter-clockwise from east (rows grow downwards): E, NE, N, NW, W, SW, S, SE
    const int step[8] = {1, -Wp + 1, -Wp, -Wp - 1, -1, Wp - 1, Wp, Wp + 1};
    int32_t mark = 1, nb = 0;
    int64_t np = 0;
    bool overflow = false;
    auto emit = [&](int64_t p) {
        if (np < cap_points) { points_xy[2 * np] = (int32_t)(p % Wp) - 1; points_xy[2 * np + 1] = (int32_t)(p / Wp) - 1; }
        else overflow = true;
        ++np;
    };
    for (int y = 1; y <= H; ++y) {
        for (int x = 1; x <= W; ++x) {
            const int64_t p0 = (int64_t)y * Wp + x;
            const int32_t v = g[p0];
            int d;
            if (v == 1 && g[p0 - 1] == 0) d = 4;              // an outer border starts here
            else if (v >= 1 && g[p0 + 1] == 0) d = 0;         // a hole border starts here
            else continue;
            ++mark;
            if (nb < cap_borders) border_start[nb] = np; else overflow = true;
            ++nb;
            int k = 0;
            while (k < 8 && g[p0 + step[(d - k + 8) & 7]] == 0) ++k;   // clockwise from the background neighbour
            if (k == 8) { g[p0] = -mark; emit(p0); continue; }        // an isolated pixel
            const int64_t p1 = p0 + step[(d - k + 8) & 7];
            int64_t prev = p1, cur = p0;
            for (;;) {
                int d0 = 0;
                for (; d0 < 8; ++d0) if (cur + step[d0] == prev) break;
                bool east_bg = false;
                int64_t nxt = cur;
                for (int q = 1; q <= 8; ++q) {                          // counter-clockwise, after the pixel we came from
                    const int dd = (d0 + q) & 7;
                    nxt = cur + step[dd];
                    if (g[nxt] != 0) break;
                    if (dd == 0) east_bg = true;
                }
                if (east_bg) g[cur] = -mark;
                else if (g[cur] == 1) g[cur] = mark;
                emit(cur);
                if (nxt == p0 && cur == p1) break;
                prev = cur; cur = nxt;
            }
        }
    }
    *n_borders = nb;
    *n_points = np;
    return overflow ? -100 : 0;          // -100: the caller's buffers were too small (the counts say how large they must be)
}
