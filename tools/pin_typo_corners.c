/* TEST TOOLING (tools/pin_typo_corners.py): the edit-distance variants the typo matcher could be mistaken for.
 * V1 = what oracle/msi_oracle.c restates (OSA; prefix mode = min over the prefixes of the dictionary word),
 * V2 = unrestricted Damerau-Levenshtein, V3 = prefix mode as "distance of the first (shortest) prefix that is within
 * the budget".  Bytes are compared (the scan is restricted to ASCII words). */
#include <stdint.h>
#include <string.h>
#define MAXL 64
static int min3(int a, int b, int c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); }

/* OSA distance of a[0..n) and b[0..m) */
int v_osa(const uint8_t *a, int n, const uint8_t *b, int m) {
  static int d[MAXL + 1][MAXL + 1];
  if (n > MAXL || m > MAXL) return 99;
  for (int i = 0; i <= n; ++i) d[i][0] = i;
  for (int j = 0; j <= m; ++j) d[0][j] = j;
  for (int i = 1; i <= n; ++i)
    for (int j = 1; j <= m; ++j) {
      int c = a[i - 1] != b[j - 1];
      d[i][j] = min3(d[i - 1][j] + 1, d[i][j - 1] + 1, d[i - 1][j - 1] + c);
      if (i > 1 && j > 1 && a[i - 1] == b[j - 2] && a[i - 2] == b[j - 1] && d[i - 2][j - 2] + 1 < d[i][j]) d[i][j] = d[i - 2][j - 2] + 1;
    }
  return d[n][m];
}
/* unrestricted Damerau-Levenshtein (Lowrance-Wagner) */
int v_damerau(const uint8_t *a, int n, const uint8_t *b, int m) {
  static int d[MAXL + 2][MAXL + 2];
  int da[256];
  if (n > MAXL || m > MAXL) return 99;
  memset(da, 0, sizeof da);
  const int inf = n + m;
  d[0][0] = inf;
  for (int i = 0; i <= n; ++i) { d[i + 1][0] = inf; d[i + 1][1] = i; }
  for (int j = 0; j <= m; ++j) { d[0][j + 1] = inf; d[1][j + 1] = j; }
  for (int i = 1; i <= n; ++i) {
    int db = 0;
    for (int j = 1; j <= m; ++j) {
      int i1 = da[b[j - 1]], j1 = db, c = 1;
      if (a[i - 1] == b[j - 1]) { c = 0; db = j; }
      int v = min3(d[i][j] + c, d[i + 1][j] + 1, d[i][j + 1] + 1);
      int t = d[i1][j1] + (i - i1 - 1) + 1 + (j - j1 - 1);
      d[i + 1][j + 1] = v < t ? v : t;
    }
    da[a[i - 1]] = i;
  }
  return d[n + 1][m + 1];
}
/* prefix mode: variant 0 = min over prefixes (V1), 1 = first prefix within `budget` (V3); dist = 0 OSA, 1 Damerau */
int v_prefix(const uint8_t *q, int n, const uint8_t *w, int m, int budget, int first, int dam) {
  int best = 99;
  for (int l = 0; l <= m; ++l) {
    int d = dam ? v_damerau(q, n, w, l) : v_osa(q, n, w, l);
    if (first) { if (d <= budget) return d; }
    else if (d < best) best = d;
  }
  return best;
}
