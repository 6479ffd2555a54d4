// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
#include "matcher_oracle.h"

#include <algorithm>
#include <climits>

namespace oracle {

// reference src/ORBmatcher.cc:1647-1663
int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
    int32_t pa[8], pb[8];
    memcpy(pa, a, 32);
    memcpy(pb, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// reference src/Frame.cc:455-470, :622-632, grid constants :160-161
void Grid::build(const FrameView& f) {
    minX = f.minX;
    minY = f.minY;
    invW = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(f.maxX - f.minX);
    invH = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(f.maxY - f.minY);
    for (int i = 0; i < FRAME_GRID_COLS; i++)
        for (int j = 0; j < FRAME_GRID_ROWS; j++) cells[i][j].clear();
    for (int i = 0; i < f.N; i++) {
        const KeyPoint& kp = f.keysUn[i];
        int posX = round((kp.x - minX) * invW);
        int posY = round((kp.y - minY) * invH);
        if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) continue;
        cells[posX][posY].push_back(i);
    }
}

// reference src/Frame.cc:567-620
std::vector<int> Grid::area(const FrameView& f, float x, float y, float r, int minLevel, int maxLevel) const {
    std::vector<int> vIndices;
    const int nMinCellX = std::max(0, (int)floor((x - minX - r) * invW));
    if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
    const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)ceil((x - minX + r) * invW));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)floor((y - minY - r) * invH));
    if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
    const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)ceil((y - minY + r) * invH));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const std::vector<int>& vCell = cells[ix][iy];
            for (size_t j = 0; j < vCell.size(); j++) {
                const KeyPoint& kpUn = f.keysUn[vCell[j]];
                if (bCheckLevels) {
                    if (kpUn.octave < minLevel) continue;
                    if (maxLevel >= 0)
                        if (kpUn.octave > maxLevel) continue;
                }
                const float distx = kpUn.x - x;
                const float disty = kpUn.y - y;
                if (fabs(distx) < r && fabs(disty) < r) vIndices.push_back(vCell[j]);
            }
        }
    }
    return vIndices;
}

// reference src/ORBmatcher.cc:1601-1642
void ComputeThreeMaxima(const int* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

// reference src/ORBmatcher.cc:45-129 (use_ratio) and :1394-1467 (best only + rotation histogram)
int SearchByProjection(const FrameView& f, const ProjQuery* q, int M, float nnratio, int use_ratio, int check_ori,
                       int* q_match, int* q_dist, int* kp_match) {
    Grid grid;
    grid.build(f);
    int nmatches = 0;
    // mvpMapPoints model: holder[k] = query index holding keypoint k, -1 = pre-existing / none
    std::vector<int> holder(f.N, -1);
    std::vector<uint8_t> blocked(f.blocked, f.blocked + f.N);
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    for (int k = 0; k < f.N; k++) kp_match[k] = -1;

    for (int i = 0; i < M; i++) {
        q_match[i] = -1;
        q_dist[i] = 256;
        const ProjQuery& p = q[i];
        if (!(p.flags & 1)) continue;
        const std::vector<int> vIndices = grid.area(f, p.u, p.v, p.radius, p.minLevel, p.maxLevel);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t c = 0; c < vIndices.size(); c++) {
            const int idx = vIndices[c];
            if (blocked[idx]) continue;   // mvpMapPoints[idx] && Observations()>0
            if (f.uRight[idx] > 0) {
                const float er = fabs(p.ur - f.uRight[idx]);
                if (er > p.radius) continue;
            }
            const int dist = DescriptorDistance(p.desc, f.desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = f.keysUn[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = f.keysUn[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (use_ratio && bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            holder[bestIdx] = i;
            blocked[bestIdx] = (p.flags & 2) ? 1 : 0;   // the new holder decides whether it blocks
            q_match[i] = bestIdx;
            q_dist[i] = bestDist;
            nmatches++;
            if (check_ori) {
                float rot = p.angle - f.keysUn[bestIdx].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx);
            }
        }
    }
    for (int k = 0; k < f.N; k++) kp_match[k] = holder[k];
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        ComputeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    kp_match[rotHist[i][j]] = -2;
                    nmatches--;
                }
            }
        }
    }
    return nmatches;
}

// cv::Mat float 3x3 * 3x1 + 3x1 is ONE cv::gemm call with flags==0 and len==3, which OpenCV 3.2 serves from
// its small-matrix branch (modules/core/src/matmul.cpp, "flags == 0 && 2 <= len && len <= 4"): the dot product is
// accumulated in float, left to right, then  d = (float)(t0*alpha + c*beta)  with alpha = beta = 1.0 (double).
// (Products with a transposed operand, e.g. -Rcw.t()*tcw, take the generic GEMMSingleMul<float,double> path.)
static inline float gemm_row(const float* a, const float* x, float c) {
    const float t0 = a[0] * x[0] + a[1] * x[1] + a[2] * x[2];
    return (float)((double)t0 * 1.0 + (double)c * 1.0);
}

// reference src/ORBmatcher.cc:1338-1392
void ProjectLastFrame(const LastFrameView& last, const float* Tcw, const float* Tlw, float fx, float fy, float cx,
                      float cy, float bf, float b, const FrameView& cur, const float* scaleFactors, float th,
                      int bMono, ProjQuery* out) {
    float Rcw[3][3], tcw[3], Rlw[3][3], tlw[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) { Rcw[r][c] = Tcw[r * 4 + c]; Rlw[r][c] = Tlw[r * 4 + c]; }
        tcw[r] = Tcw[r * 4 + 3];
        tlw[r] = Tlw[r * 4 + 3];
    }
    // twc = -Rcw.t()*tcw ; tlc = Rlw*twc + tlw
    float twc[3], tlc[3];
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Rcw[k][r] * (double)tcw[k];
        twc[r] = (float)(-1.0 * s);
    }
    for (int r = 0; r < 3; r++) tlc[r] = gemm_row(Rlw[r], twc, tlw[r]);
    const bool bForward = tlc[2] > b && !bMono;
    const bool bBackward = -tlc[2] > b && !bMono;

    for (int i = 0; i < last.N; i++) {
        ProjQuery& q = out[i];
        memset(&q, 0, sizeof(q));
        q.minLevel = -1;
        q.maxLevel = -1;
        if (!(last.has_mp[i] & 1)) continue;
        const float* X = last.Xw + 3 * i;
        const float xc = gemm_row(Rcw[0], X, tcw[0]);
        const float yc = gemm_row(Rcw[1], X, tcw[1]);
        const float zc = gemm_row(Rcw[2], X, tcw[2]);
        const float invzc = 1.0 / zc;
        if (invzc < 0) continue;
        float u = fx * xc * invzc + cx;
        float v = fy * yc * invzc + cy;
        if (u < cur.minX || u > cur.maxX) continue;
        if (v < cur.minY || v > cur.maxY) continue;
        const int nLastOctave = last.keys[i].octave;
        const float radius = th * scaleFactors[nLastOctave];
        if (bForward) { q.minLevel = nLastOctave; q.maxLevel = -1; }
        else if (bBackward) { q.minLevel = 0; q.maxLevel = nLastOctave; }
        else { q.minLevel = nLastOctave - 1; q.maxLevel = nLastOctave + 1; }
        q.u = u;
        q.v = v;
        q.ur = u - bf * invzc;
        q.radius = radius;
        q.flags = 1 | ((last.has_mp[i] & 2) ? 2 : 0);
        q.angle = last.keys[i].angle;
        memcpy(q.desc, last.mp_desc + (size_t)i * 32, 32);
    }
}

}  // namespace oracle

namespace oracle {
// reference src/Frame.cc:706-880
void ComputeStereoMatches(int N, const KeyPoint* keysL, const uint8_t* descL, int Nr, const KeyPoint* keysR,
                          const uint8_t* descR, const std::vector<Image>& pyrL, const std::vector<Image>& pyrR,
                          const float* mvScaleFactors, const float* mvInvScaleFactors, float mbf, float mb, float* mvuRight,
                          float* mvDepth) {
    for (int i = 0; i < N; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = pyrL[0].h;
    std::vector<std::vector<size_t>> vRowIndices(nRows);
    for (int iR = 0; iR < Nr; iR++) {
        const KeyPoint& kp = keysR[iR];
        const float kpY = kp.y;
        const float r = 2.0f * mvScaleFactors[keysR[iR].octave];
        const int maxr = ceil(kpY + r);
        const int minr = floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);
    }
    const float minZ = mb;
    const float minD = 0;
    const float maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < N; iL++) {
        const KeyPoint& kpL = keysL[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y;
        const float uL = kpL.x;
        if ((int)vL < 0 || (int)vL >= nRows) continue;
        const std::vector<size_t>& vCandidates = vRowIndices[(int)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD;
        const float maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t* dL = descL + (size_t)iL * 32;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const KeyPoint& kpR = keysR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = DescriptorDistance(dL, descR + iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = keysR[bestIdxR].x;
            const float scaleFactor = mvInvScaleFactors[kpL.octave];
            const float scaleduL = round(kpL.x * scaleFactor);
            const float scaledvL = round(kpL.y * scaleFactor);
            const float scaleduR0 = round(uR0 * scaleFactor);
            const int w = 5;
            const Image& imL = pyrL[kpL.octave];
            const Image& imR = pyrR[kpL.octave];
            const int cy = (int)scaledvL, cxL = (int)scaleduL;
            const float centerL = imL.row(cy)[cxL];
            int bestDist2 = INT32_MAX;
            int bestincR = 0;
            const int L = 5;
            std::vector<float> vDists(2 * L + 1);
            const float iniu = scaleduR0 + L - w;
            const float endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= imR.w) continue;
            // The reference reads the left 11x11 window and the right windows [uR0-10, uR0+10] without
            // checking them (its :810-811 test covers only part of the right side).  Skip where that
            // would leave the image (undefined behaviour in the reference).
            if (cy - w < 0 || cy + w >= imL.h || cxL - w < 0 || cxL + w >= imL.w || (int)scaleduR0 - L - w < 0 ||
                (int)scaleduR0 + L + w >= imR.w)
                continue;
            for (int incR = -L; incR <= +L; incR++) {
                const int cxR = (int)(scaleduR0 + incR);
                const float centerR = imR.row(cy)[cxR];
                double acc = 0;   // cv::norm(NORM_L1) on CV_32F accumulates in double
                for (int dy = -w; dy <= w; dy++)
                    for (int dx = -w; dx <= w; dx++) {
                        const float a = (float)imL.row(cy + dy)[cxL + dx] - centerL;
                        const float bb = (float)imR.row(cy + dy)[cxR + dx] - centerR;
                        acc += std::fabs(a - bb);
                    }
                float dist = (float)acc;
                if (dist < bestDist2) { bestDist2 = dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1];
            const float dist2 = vDists[L + bestincR];
            const float dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = mvScaleFactors[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDist2, iL));
            }
        }
    }
    if (vDistIdx.empty()) return;
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        mvuRight[vDistIdx[i].second] = -1;
        mvDepth[vDistIdx[i].second] = -1;
    }
}
}  // namespace oracle

namespace oracle {
// reference src/ORBmatcher.cc:888-947
int FuseSearch(const FrameView& f, const ProjQuery* q, int M, const float* invLevelSigma2, int* q_match, int* q_dist) {
    Grid grid;
    grid.build(f);
    int n = 0;
    for (int i = 0; i < M; i++) {
        q_match[i] = -1;
        q_dist[i] = 256;
        const ProjQuery& p = q[i];
        if (!(p.flags & 1)) continue;
        const float u = p.u, v = p.v, ur = p.ur;
        const std::vector<int> vIndices = grid.area(f, u, v, p.radius, -1, -1);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestIdx = -1;
        for (size_t c = 0; c < vIndices.size(); c++) {
            const int idx = vIndices[c];
            const KeyPoint& kp = f.keysUn[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < p.minLevel || kpLevel > p.maxLevel) continue;
            if (f.uRight[idx] >= 0) {
                const float ex = u - kp.x, ey = v - kp.y, er = ur - f.uRight[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * invLevelSigma2[kpLevel] > 7.8) continue;
            } else {
                const float ex = u - kp.x, ey = v - kp.y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * invLevelSigma2[kpLevel] > 5.99) continue;
            }
            const int dist = DescriptorDistance(p.desc, f.desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { q_match[i] = bestIdx; q_dist[i] = bestDist; n++; }
    }
    return n;
}
}  // namespace oracle

namespace oracle {
static int find_node(const BowSide2& s2, uint32_t node) {
    const uint32_t* e = s2.nodes + s2.nNodes;
    const uint32_t* p = std::lower_bound(s2.nodes, e, node);
    return (p != e && *p == node) ? (int)(p - s2.nodes) : -1;
}

// reference src/ORBmatcher.cc:159-288
int SearchByBoW(int nq, const int32_t* q_idx1, const uint32_t* q_node, const KeyPoint* keys1, const uint8_t* desc1,
                const uint8_t* valid1, int N2, const KeyPoint* keys2, const uint8_t* desc2, const BowSide2& s2, float nnratio,
                int checkOri, int* match_f) {
    for (int k = 0; k < N2; k++) match_f[k] = -1;
    int nmatches = 0;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    for (int q = 0; q < nq; q++) {
        const int realIdxKF = q_idx1[q];
        if (!valid1[realIdxKF]) continue;
        const int nd = find_node(s2, q_node[q]);
        if (nd < 0) continue;
        const uint8_t* dKF = desc1 + (size_t)realIdxKF * 32;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int t = s2.start[nd]; t < s2.start[nd + 1]; t++) {
            const int realIdxF = s2.items[t];
            if (match_f[realIdxF] >= 0) continue;
            const int dist = DescriptorDistance(dKF, desc2 + (size_t)realIdxF * 32);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 <= TH_LOW) {
            if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                match_f[bestIdxF] = realIdxKF;
                if (checkOri) {
                    float rot = keys1[realIdxKF].angle - keys2[bestIdxF].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(bestIdxF);
                }
                nmatches++;
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        ComputeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { match_f[rotHist[i][j]] = -2; nmatches--; }
        }
    }
    return nmatches;
}

// reference src/ORBmatcher.cc:140-157
static bool CheckDistEpipolarLine(const KeyPoint& kp1, const KeyPoint& kp2, const float* F12, const float* levelSigma2) {
    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * levelSigma2[kp2.octave];
}

// reference src/ORBmatcher.cc:657-823
int SearchForTriangulation(int nq, const int32_t* q_idx1, const uint32_t* q_node, int N1, const KeyPoint* keys1,
                           const uint8_t* desc1, const float* uRight1, const uint8_t* skip1, int N2, const KeyPoint* keys2,
                           const uint8_t* desc2, const float* uRight2, const uint8_t* has_mp2, const BowSide2& s2,
                           const float* F12, float ex, float ey, const float* scaleFactors, const float* levelSigma2,
                           int bOnlyStereo, int checkOri, int* match12) {
    for (int i = 0; i < N1; i++) match12[i] = -1;
    int nmatches = 0;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    for (int q = 0; q < nq; q++) {
        const int idx1 = q_idx1[q];
        if (skip1[idx1]) continue;
        const bool bStereo1 = uRight1[idx1] >= 0;
        if (bOnlyStereo && !bStereo1) continue;
        const int nd = find_node(s2, q_node[q]);
        if (nd < 0) continue;
        const KeyPoint& kp1 = keys1[idx1];
        const uint8_t* d1 = desc1 + (size_t)idx1 * 32;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int t = s2.start[nd]; t < s2.start[nd + 1]; t++) {
            const int idx2 = s2.items[t];
            if (has_mp2[idx2]) continue;   // vbMatched2 is never set in the reference
            const bool bStereo2 = uRight2[idx2] >= 0;
            if (bOnlyStereo && !bStereo2) continue;
            const int dist = DescriptorDistance(d1, desc2 + (size_t)idx2 * 32);
            if (dist > TH_LOW || dist > bestDist) continue;
            const KeyPoint& kp2 = keys2[idx2];
            if (!bStereo1 && !bStereo2) {
                const float distex = ex - kp2.x, distey = ey - kp2.y;
                if (distex * distex + distey * distey < 100 * scaleFactors[kp2.octave]) continue;
            }
            if (CheckDistEpipolarLine(kp1, kp2, F12, levelSigma2)) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
            match12[idx1] = bestIdx2;
            nmatches++;
            if (checkOri) {
                float rot = kp1.angle - keys2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(idx1);
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        ComputeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { match12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}
}  // namespace oracle
