// ORACLE (test infrastructure): C entry point around the REFERENCE's own C++ decode/NMS code
// (examples/YOLO-Master-Cross-Platform-Edge-Deployment/cpp/src/common.cpp, compiled in place by build.py).
// Nothing of the algorithm lives here: the wrapper only marshals arrays into the reference's structs.
#include "yolomaster.hpp"

extern "C" int cwref_nms_and_cap(const float* xywh /* [n][4] top-left x, y, w, h */, const float* score, const int* cls, int n,
                                 float conf, float iou, int max_det, int cluster_weighted, float sigma, int orig_w,
                                 int orig_h, float* out /* [max_det][6]: x, y, w, h, conf, class */) {
    std::vector<yolomaster::RawDet> cands((size_t)n);
    for (int i = 0; i < n; ++i) {
        cands[i].box = cv::Rect2f(xywh[4 * i], xywh[4 * i + 1], xywh[4 * i + 2], xywh[4 * i + 3]);
        cands[i].score = score[i];
        cands[i].cls = cls[i];
    }
    yolomaster::Config cfg;
    cfg.conf_thresh = conf;
    cfg.iou_thresh = iou;
    cfg.max_det = max_det;
    cfg.nms_mode = cluster_weighted ? yolomaster::NmsMode::ClusterWeighted : yolomaster::NmsMode::Standard;
    cfg.cw_sigma = sigma;
    const std::vector<yolomaster::Detection> d = yolomaster::nms_and_cap(cands, cfg, orig_w, orig_h);
    for (size_t i = 0; i < d.size(); ++i) {
        float* o = out + 6 * i;
        o[0] = d[i].box.x; o[1] = d[i].box.y; o[2] = d[i].box.width; o[3] = d[i].box.height;
        o[4] = d[i].conf; o[5] = (float)d[i].class_id;
    }
    return (int)d.size();
}
