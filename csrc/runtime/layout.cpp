// Page layout analysis in native code: recursive XY-cut reading order over the line boxes (and figure boxes) of a
// page, with running header / footer detection.  Same algorithm, same thresholds and the same results as
// distributed_llms_amd/runtime/layout.py, which documents the method and serves as the oracle in the tests; the
// engine calls this version because it orders every page on its one host thread (tens of microseconds here against
// milliseconds in numpy).
//
// Role in the pipeline: what magic-pdf's layoutreader / xy-cut sort does inside pipe_ocr_mode / pipe_txt_mode in the
// reference (services/ocr_service/ocr_service.py:215-220, download_models_hf.py:51).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

namespace {

struct Box { double x0, y0, x1, y1; };
struct Gap { double width, start, end; };

struct Cutter {
  const std::vector<Box>& b;
  double h_med, tx, ty, text_aspect;
  int min_col_lines;
  std::vector<int> order;
  std::vector<int> block_of;
  std::vector<char> table_block;   // per leaf: its boxes are the cells of a table, in row-major order
  int n_blocks = 0;

  Cutter(const std::vector<Box>& boxes, double col_gap, double block_gap, int mcl, double ta)
      : b(boxes), text_aspect(ta), min_col_lines(mcl), block_of(boxes.size(), 0) {
    std::vector<double> h;
    for (auto& x : b) h.push_back(x.y1 - x.y0);
    h_med = median(h);
    tx = col_gap * h_med;
    ty = block_gap * h_med;
  }

  static double median(std::vector<double> v) {
    if (v.empty()) return 1.0;
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    return n % 2 ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]);
  }

  template <bool Y>
  std::vector<Gap> gaps(const std::vector<int>& idx) const {
    std::vector<int> o(idx);
    std::stable_sort(o.begin(), o.end(), [&](int p, int q) { return (Y ? b[p].y0 : b[p].x0) < (Y ? b[q].y0 : b[q].x0); });
    std::vector<Gap> out;
    double reach = Y ? b[o[0]].y1 : b[o[0]].x1;
    for (size_t k = 1; k < o.size(); ++k) {
      const double lo = Y ? b[o[k]].y0 : b[o[k]].x0, hi = Y ? b[o[k]].y1 : b[o[k]].x1;
      if (lo > reach) out.push_back({lo - reach, reach, lo});
      reach = std::max(reach, hi);
    }
    return out;
  }

  template <bool Y>
  std::vector<std::vector<int>> split_at(const std::vector<int>& idx, std::vector<double> cuts) const {
    std::sort(cuts.begin(), cuts.end());
    std::vector<std::vector<int>> parts(cuts.size() + 1);
    for (int i : idx) {
      const double lo = (Y ? b[i].y0 : b[i].x0) + 1e-6;
      const size_t s = std::lower_bound(cuts.begin(), cuts.end(), lo) - cuts.begin();
      parts[s].push_back(i);
    }
    std::vector<std::vector<int>> out;
    for (auto& p : parts) if (!p.empty()) out.push_back(std::move(p));
    return out;
  }

  // Rows of a region: boxes whose vertical centres lie within half a line height of the row's first box, each row
  // left to right (the order ``leaf`` emits).
  std::vector<std::vector<int>> rows_of(const std::vector<int>& idx) const {
    std::vector<int> o(idx);
    auto cy = [&](int i) { return 0.5 * (b[i].y0 + b[i].y1); };
    std::stable_sort(o.begin(), o.end(), [&](int p, int q) { return cy(p) < cy(q); });
    std::vector<std::vector<int>> rows;
    size_t start = 0;
    for (size_t i = 1; i <= o.size(); ++i) {
      if (i == o.size() || cy(o[i]) - cy(o[start]) > 0.5 * h_med) {
        std::vector<int> r(o.begin() + start, o.begin() + i);
        std::stable_sort(r.begin(), r.end(), [&](int p, int q) { return b[p].x0 < b[q].x0; });
        rows.push_back(std::move(r));
        start = i;
      }
    }
    return rows;
  }

  // A region is a table when its boxes form a grid: at least ``min_rows`` rows of two or more cells each, the cells
  // of all rows falling into common columns (the whitespace between the columns runs through every row, at least 0.6
  // line heights wide), no row with two cells in one column, every column filled in at least half of the rows, and at
  // most one column of long text (two columns of running text whose lines happen to share baselines are not a table).
  // Cells may wrap: a row of fewer boxes at line spacing under a table row (``continues_row``) belongs to that row.
  // Column boundaries of a would-be table: the middles of the x-intervals, at least 0.6 line heights wide, that at
  // most ``tol`` cells reach into (one over-long cell that runs up to its neighbour must not glue two columns of an
  // eight-row table together), with cells on both sides.
  std::vector<double> table_cuts(const std::vector<int>& idx, int tol) const {
    std::vector<double> edges;
    for (int i : idx) { edges.push_back(b[i].x0); edges.push_back(b[i].x1); }
    std::sort(edges.begin(), edges.end());
    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
    std::vector<double> cuts;
    bool open = false;
    double start = 0, end = 0;
    for (size_t k = 0; k + 1 < edges.size(); ++k) {
      const double a = edges[k], e = edges[k + 1];
      int cover = 0;
      for (int i : idx) cover += (b[i].x0 < e && b[i].x1 > a);
      if (cover <= tol) {
        if (!open) { open = true; start = a; }
        end = e;
      } else {
        if (open && end - start >= 0.6 * h_med) {
          bool left = false;
          for (int i : idx) left |= b[i].x1 <= start + 1e-9;
          if (left) cuts.push_back(0.5 * (start + end));
        }
        open = false;
      }
    }
    return cuts;
  }

  bool is_table(const std::vector<int>& idx, int min_rows) const {
    if ((int)idx.size() < 2 * min_rows) return false;
    for (int i : idx) if (b[i].y1 - b[i].y0 > 3.0 * h_med) return false;     // a figure or a rotated line: no cell
    auto rows = rows_of(idx);
    int n_full = 0;
    for (auto& r : rows) n_full += r.size() >= 2;
    if (n_full < min_rows || rows[0].size() < 2) return false;
    std::vector<double> cuts = table_cuts(idx, n_full >= 6 ? 1 : 0);
    if (cuts.empty()) return false;
    const size_t C = cuts.size() + 1;
    std::vector<int> filled(C, 0);
    std::vector<std::vector<double>> aspect(C);
    int n_rows = 0;
    std::vector<char> base_cols(C, 0);
    size_t base_size = 0;
    // the white between the cells of two complete rows that follow each other, column by column: the lines of one cell
    // are closer than that (in two columns of short lines that merely share some baselines they are not)
    size_t most = 0;
    for (auto& r : rows) most = std::max(most, r.size());
    std::vector<std::vector<size_t>> cols_of(rows.size());
    for (size_t k = 0; k < rows.size(); ++k)
      for (int i : rows[k]) cols_of[k].push_back(std::lower_bound(cuts.begin(), cuts.end(), b[i].x0 + 1e-6) - cuts.begin());
    std::vector<double> row_gaps;
    for (size_t k = 1; k < rows.size(); ++k)
      if (rows[k].size() == most && rows[k - 1].size() == most && cols_of[k] == cols_of[k - 1])
        for (size_t t = 0; t < rows[k].size(); ++t) row_gaps.push_back(b[rows[k][t]].y0 - b[rows[k - 1][t]].y1);
    const bool has_ref = !row_gaps.empty();
    const double ref_gap = has_ref ? median(row_gaps) : 0.0;
    std::vector<double> bottom(C, 0.0);                  // lower edge of the last box seen in each column
    for (size_t k = 0; k < rows.size(); ++k) {
      auto& r = rows[k];
      const auto& col = cols_of[k];
      std::vector<char> seen(C, 0);
      for (size_t c : col) {
        if (seen[c]) return false;
        seen[c] = 1;
      }
      // the second (third ...) line of cells that wrap: part of the row above, no row of its own
      const bool cont = k > 0 && continues_row(r, col, bottom, base_cols, base_size, cuts, has_ref, ref_gap);
      if (!cont) {
        if (r.size() < 2) return false;
        ++n_rows;
        base_cols = seen;
        base_size = r.size();
      }
      for (size_t t = 0; t < r.size(); ++t) {
        const int i = r[t];
        if (!cont) ++filled[col[t]];
        aspect[col[t]].push_back((b[i].x1 - b[i].x0) / std::max(b[i].y1 - b[i].y0, 1.0));
        bottom[col[t]] = b[i].y1;
      }
    }
    if (n_rows < min_rows) return false;
    int wide = 0;
    for (size_t c = 0; c < C; ++c) {
      if (2 * filled[c] < n_rows) return false;
      if (median(aspect[c]) >= 10.0) ++wide;
    }
    return wide <= 1;
  }

  // White between a row and the row above it (negative when their boxes overlap).
  double row_gap(const std::vector<int>& r, const std::vector<int>& above) const {
    double top = 1e300, bottom = -1e300;
    for (int i : r) top = std::min(top, b[i].y0);
    for (int i : above) bottom = std::max(bottom, b[i].y1);
    return top - bottom;
  }

  // Is row ``r`` the next line of cells of the table row above it that wrap?  It has fewer boxes than that row has
  // cells, each in a column the row has a cell in and none reaching over a column boundary, and each follows the box
  // above it in its column (``bottom``) at line spacing: less than three quarters of the white between the cells of
  // two complete rows of this table (``ref_gap``) and at most a line height; at most 0.6 line heights where the table
  // has no two complete rows in a row to compare with.
  bool continues_row(const std::vector<int>& r, const std::vector<size_t>& col, const std::vector<double>& bottom,
                     const std::vector<char>& base_cols, size_t base_size, const std::vector<double>& cuts,
                     bool has_ref, double ref_gap) const {
    if (r.size() >= base_size || base_size == 0) return false;
    for (size_t t = 0; t < r.size(); ++t) {
      if (!base_cols[col[t]]) return false;
      if (col[t] < cuts.size() && b[r[t]].x1 > cuts[col[t]] + 0.1 * h_med) return false;
      const double gap = b[r[t]].y0 - bottom[col[t]];
      if (gap > (has_ref ? 1.0 : 0.6) * h_med || (has_ref && gap >= 0.75 * ref_gap)) return false;
    }
    return true;
  }

  void leaf(const std::vector<int>& idx, bool table = false) {
    table_block.push_back(table ? 1 : 0);
    std::vector<int> o(idx);
    auto cy = [&](int i) { return 0.5 * (b[i].y0 + b[i].y1); };
    std::stable_sort(o.begin(), o.end(), [&](int p, int q) { return cy(p) < cy(q); });
    size_t start = 0;
    for (size_t i = 1; i <= o.size(); ++i) {
      if (i == o.size() || cy(o[i]) - cy(o[start]) > 0.5 * h_med) {
        std::stable_sort(o.begin() + start, o.begin() + i, [&](int p, int q) { return b[p].x0 < b[q].x0; });
        start = i;
      }
    }
    for (int i : o) { order.push_back(i); block_of[i] = n_blocks; }
    ++n_blocks;
  }

  // A side of a vertical cut counts as a column of running text when it holds several lines that are much wider than
  // tall: the median aspect reaches text_aspect, or - a column that also holds a table, whose many short cells pull
  // the median down - at least min_col_lines of its boxes are long lines (aspect >= 8; a column of a table has none).
  bool is_text_column(const std::vector<int>& idx) const {
    if ((int)idx.size() < min_col_lines) return false;
    std::vector<double> r;
    int longs = 0;
    for (int i : idx) {
      r.push_back((b[i].x1 - b[i].x0) / std::max(b[i].y1 - b[i].y0, 1.0));
      longs += r.back() >= 8.0;
    }
    if (median(r) >= text_aspect) return true;
    return longs >= min_col_lines;
  }

  std::vector<double> column_cuts(const std::vector<int>& idx) const {
    std::vector<double> cuts;
    for (auto& g : gaps<false>(idx)) if (g.width >= tx) cuts.push_back(0.5 * (g.start + g.end));
    if (cuts.empty()) return cuts;
    auto parts = split_at<false>(idx, cuts);
    if (parts.size() < 2) return {};
    for (auto& p : parts) if (!is_text_column(p)) return {};
    return cuts;
  }

  void cut(const std::vector<int>& idx, int depth) {
    if (idx.empty()) return;
    if (idx.size() == 1 || depth > 12) { leaf(idx); return; }
    // 0. a grid of cells is read row by row, whatever its columns look like
    if (is_table(idx, 3)) { leaf(idx, true); return; }
    auto cuts = column_cuts(idx);
    if (!cuts.empty()) {
      for (auto& part : split_at<false>(idx, cuts)) cut(part, depth + 1);
      return;
    }
    std::vector<double> ycuts;
    for (auto& g : gaps<true>(idx)) if (g.width >= ty) ycuts.push_back(0.5 * (g.start + g.end));
    if (!ycuts.empty()) {
      auto parts = split_at<true>(idx, ycuts);
      if (parts.size() > 1) {
        size_t i = 0;
        while (i < parts.size()) {
          size_t j = i;
          std::vector<int> merged(parts[i]);
          // the rows of a table whose row spacing is as large as a block gap belong together
          while (j + 1 < parts.size()) {
            std::vector<int> trial(merged);
            trial.insert(trial.end(), parts[j + 1].begin(), parts[j + 1].end());
            if (!is_table(trial, 2)) break;
            merged.swap(trial);
            ++j;
          }
          if (j > i && is_table(merged, 3)) {
            leaf(merged, true);
            i = j + 1;
            continue;
          }
          j = i;
          merged = parts[i];
          while (j + 1 < parts.size()) {
            std::vector<int> trial(merged);
            trial.insert(trial.end(), parts[j + 1].begin(), parts[j + 1].end());
            if (column_cuts(trial).empty()) break;
            merged.swap(trial);
            ++j;
          }
          cut(merged, depth + 1);
          i = j + 1;
        }
        return;
      }
    }
    double x0 = 1e300, x1 = -1e300;
    for (int i : idx) { x0 = std::min(x0, b[i].x0); x1 = std::max(x1, b[i].x1); }
    std::vector<int> wide, rest;
    for (int i : idx) ((b[i].x1 - b[i].x0) >= 0.6 * (x1 - x0) ? wide : rest).push_back(i);
    if (!wide.empty() && (int)rest.size() >= 2 * min_col_lines && wide.size() < idx.size() && !column_cuts(rest).empty()) {
      std::stable_sort(wide.begin(), wide.end(), [&](int p, int q) { return b[p].y0 < b[q].y0; });
      std::vector<double> ys;
      for (int k : wide) ys.push_back(0.5 * (b[k].y0 + b[k].y1));
      std::vector<std::vector<int>> bands(ys.size() + 1);
      for (int i : rest) {
        const double c = 0.5 * (b[i].y0 + b[i].y1);
        bands[std::lower_bound(ys.begin(), ys.end(), c) - ys.begin()].push_back(i);
      }
      for (size_t s = 0; s <= ys.size(); ++s) {
        cut(bands[s], depth + 1);
        if (s < ys.size()) leaf({wide[s]});
      }
      return;
    }
    leaf_with_tables(idx);
  }

  // A region that no cut divides any further may still hold a table between its paragraphs (a table set close under a
  // paragraph leaves no block gap): consecutive rows of two or more cells that form a grid become a table leaf, what
  // lies above and below stays ordinary leaves, top to bottom.
  // Last pass over the finished reading sequence: rows of a table that the cuts left in different leaves (the first
  // row glued to the paragraph above it, the others a block gap apart) are consecutive in the sequence all the same.
  // Consecutive rows of two or more cells, at most three line heights from one to the next, that form a grid become
  // one table block of their own.
  void tables_across_leaves() {
    const size_t n = order.size();
    auto cy = [&](int i) { return 0.5 * (b[i].y0 + b[i].y1); };
    std::vector<std::pair<size_t, size_t>> rows;           // [first, last + 1) positions in order
    size_t s = 0;
    for (size_t k = 1; k <= n; ++k) {
      if (k == n || std::fabs(cy(order[k]) - cy(order[s])) > 0.5 * h_med || b[order[k]].x0 <= b[order[k - 1]].x0) {
        rows.push_back({s, k});
        s = k;
      }
    }
    size_t i = 0;
    while (i < rows.size()) {
      auto multi = [&](size_t r) { return rows[r].second - rows[r].first >= 2; };
      if (!multi(i)) { ++i; continue; }
      auto follows = [&](size_t r, double lim) {
        return b[order[rows[r].first]].y0 - b[order[rows[r - 1].first]].y1 <= lim * h_med &&
               cy(order[rows[r].first]) > cy(order[rows[r - 1].first]);
      };
      size_t j = i + 1;
      while (j < rows.size() && multi(j) && follows(j, 3.0)) ++j;
      // the same run with the further lines of wrapped cells (single boxes at line spacing) counted in; tried first,
      // the run of complete rows alone after it
      size_t e = i + 1;
      int full = 1;
      while (e < rows.size() && follows(e, multi(e) ? 3.0 : 1.0)) { full += multi(e); ++e; }
      const std::pair<size_t, int> tries[2] = {{e, full}, {j, (int)(j - i)}};
      for (int t = (e > j ? 0 : 1); t < 2; ++t) {
        const size_t end = tries[t].first;
        if (tries[t].second < 3) continue;
        bool all_table = true;
        for (size_t k = rows[i].first; k < rows[end - 1].second; ++k) all_table &= table_block[block_of[order[k]]] != 0;
        if (all_table) continue;
        std::vector<int> cells(order.begin() + rows[i].first, order.begin() + rows[end - 1].second);
        if (is_table(cells, 3)) {
          table_block.push_back(1);
          for (int c : cells) block_of[c] = n_blocks;
          ++n_blocks;
          j = end;
          break;
        }
      }
      i = j;
    }
  }

  void leaf_with_tables(const std::vector<int>& idx) {
    auto rows = rows_of(idx);
    const size_t R = rows.size();
    std::vector<std::pair<size_t, size_t>> tables;       // [first row, last row + 1)
    size_t i = 0;
    while (i < R) {
      if (rows[i].size() < 2) { ++i; continue; }
      size_t j = i;
      while (j < R && rows[j].size() >= 2) ++j;
      // with the lines of wrapped cells between its rows a table is a longer run: rows of fewer boxes that follow the
      // row above at line spacing count in (what comes of it is checked as a whole, then without them)
      auto tight = [&](size_t e) { return row_gap(rows[e], rows[e - 1]) <= 1.0 * h_med; };
      size_t e = i;
      while (e < R && (rows[e].size() >= 2 || (e > i && tight(e)))) ++e;
      if (e > j) {
        int full = 0;
        std::vector<int> cells;
        for (size_t r = i; r < e; ++r) { full += rows[r].size() >= 2; cells.insert(cells.end(), rows[r].begin(), rows[r].end()); }
        if (full >= 3 && is_table(cells, 3)) { tables.push_back({i, e}); i = e; continue; }
      }
      if (j - i >= 3) {
        std::vector<int> cells;
        for (size_t r = i; r < j; ++r) cells.insert(cells.end(), rows[r].begin(), rows[r].end());
        if (is_table(cells, 3)) tables.push_back({i, j});
      }
      i = j;
    }
    if (tables.empty()) { leaf(idx); return; }
    size_t at = 0;
    auto emit = [&](size_t a, size_t b, bool table) {
      if (a >= b) return;
      std::vector<int> part;
      for (size_t r = a; r < b; ++r) part.insert(part.end(), rows[r].begin(), rows[r].end());
      leaf(part, table);
    };
    for (auto& t : tables) {
      emit(at, t.first, false);
      emit(t.first, t.second, true);
      at = t.second;
    }
    emit(at, R, false);
  }
};

void reading_order(const std::vector<Box>& boxes, std::vector<int>* order, std::vector<int>* block,
                   std::vector<char>* in_table = nullptr) {
  order->clear();
  block->assign(boxes.size(), 0);
  if (in_table) in_table->assign(boxes.size(), 0);
  if (boxes.empty()) return;
  Cutter c(boxes, 1.2, 0.9, 3, 4.0);
  std::vector<int> all(boxes.size());
  std::iota(all.begin(), all.end(), 0);
  c.cut(all, 0);
  c.tables_across_leaves();
  *order = c.order;
  *block = c.block_of;
  if (in_table)
    for (size_t i = 0; i < boxes.size(); ++i) (*in_table)[i] = c.table_block[c.block_of[i]];
}

}  // namespace

extern "C" {

// boxes [n][4], figures [nf][4] (x0, y0, x1, y1, y down).  Writes n + nf entries each of order_out (indices into
// [boxes ; figures]), block_out and kind_out (0 text line, 1 figure, 2 cell of a table: the cells of one table share a
// block and come row by row), aligned with order_out.  Returns n + nf.
int dlrt_layout_order(const float* boxes, int n, const float* figs, int nf, float page_h, int* order_out, int* block_out,
                      int* kind_out) {
  if (n < 0 || nf < 0) return -1;
  std::vector<Box> bx(n);
  for (int i = 0; i < n; ++i) bx[i] = {boxes[4 * i], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3]};
  // running header / footer: entirely inside the top / bottom 7.5 % of the page, a line height away from the body
  std::vector<int> head, body, foot;
  const double band = 0.075;
  if (n < 3 || page_h <= 0) {
    for (int i = 0; i < n; ++i) body.push_back(i);
  } else {
    std::vector<double> h;
    for (auto& x : bx) h.push_back(x.y1 - x.y0);
    const double h_med = Cutter::median(h);
    std::vector<char> top(n), bot(n);
    bool any_body = false, any_top = false, any_bot = false;
    for (int i = 0; i < n; ++i) {
      top[i] = bx[i].y1 <= band * page_h;
      bot[i] = bx[i].y0 >= (1.0 - band) * page_h;
      any_body |= !(top[i] || bot[i]);
      any_top |= top[i];
      any_bot |= bot[i];
    }
    if (!any_body) {
      for (int i = 0; i < n; ++i) body.push_back(i);
    } else {
      double body_y0 = 1e300, top_y1 = -1e300;
      for (int i = 0; i < n; ++i) {
        if (!(top[i] || bot[i])) body_y0 = std::min(body_y0, bx[i].y0);
        if (top[i]) top_y1 = std::max(top_y1, bx[i].y1);
      }
      if (any_top && body_y0 - top_y1 < h_med) std::fill(top.begin(), top.end(), 0);
      double body_y1 = -1e300, bot_y0 = 1e300;
      for (int i = 0; i < n; ++i) {
        if (!(top[i] || bot[i])) body_y1 = std::max(body_y1, bx[i].y1);
        if (bot[i]) bot_y0 = std::min(bot_y0, bx[i].y0);
      }
      if (any_bot && bot_y0 - body_y1 < h_med) {
        // the footer joins the body; the body's lower edge is unchanged for the header test above (already applied)
        std::fill(bot.begin(), bot.end(), 0);
      }
      for (int i = 0; i < n; ++i) (top[i] ? head : bot[i] ? foot : body).push_back(i);
    }
  }
  int k = 0, nblk = 0;
  std::vector<int> o, blk;
  auto emit_group = [&](const std::vector<int>& grp) {
    std::vector<Box> g;
    for (int i : grp) g.push_back(bx[i]);
    reading_order(g, &o, &blk);
    for (int i : o) { order_out[k] = grp[i]; block_out[k] = nblk; kind_out[k] = 0; ++k; }
    if (!o.empty()) ++nblk;
  };
  emit_group(head);
  {
    std::vector<Box> g;
    for (int i : body) g.push_back(bx[i]);
    for (int i = 0; i < nf; ++i) g.push_back({figs[4 * i], figs[4 * i + 1], figs[4 * i + 2], figs[4 * i + 3]});
    std::vector<char> in_table;
    reading_order(g, &o, &blk, &in_table);
    int mx = -1;
    for (int i : o) {
      const bool fig = i >= (int)body.size();
      order_out[k] = fig ? n + (i - (int)body.size()) : body[i];
      block_out[k] = nblk + blk[i];
      kind_out[k] = fig ? 1 : (in_table[i] ? 2 : 0);
      mx = std::max(mx, blk[i]);
      ++k;
    }
    nblk += mx + 1;
  }
  {
    std::vector<Box> g;
    for (int i : foot) g.push_back(bx[i]);
    reading_order(g, &o, &blk);
    for (int i : o) { order_out[k] = foot[i]; block_out[k] = nblk; kind_out[k] = 0; ++k; }
  }
  return k;
}

}  // extern "C"
