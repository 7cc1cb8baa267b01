/* fga_spec.c -- alignment specification tables (host): replaces New_Align_Spec / set_table / Bias_Factor
 * (reference align.c:178-268).  PATH_AVE and the two 32768-entry int16 tables over 15-bit match patterns:
 *   table[p] = (score of p) - (max prefix score of p),  score[p] = score of p,
 * with match = +mscore, mismatch = -dscore, mscore = int(1000 * bias * (1-ave_corr)), dscore = 1000 - mscore.
 * The base-composition bias comes from freq[0]+freq[3] (the GDB's 'f' line).
 */
#include <stdint.h>
#include <stdlib.h>

#include "fga_host.h"
#include "fastga_amd.h"

#define TRIM_LEN 15
#define PATH_LEN 60

static void fill(int bit, int prefix, int score, int max, int ms, int ds, int16_t *table, int16_t *sc)
{ if (bit >= TRIM_LEN)
    { table[prefix] = (int16_t) (score-max);
      sc[prefix]    = (int16_t) score;
      return;
    }
  if (score > max)
    max = score;
  fill(bit+1,prefix<<1,score-ds,max,ms,ds,table,sc);
  fill(bit+1,(prefix<<1)|1,score+ms,max,ms,ds,table,sc);
}

int fga_align_spec(double ave_corr, int tspace, const float *freq, int *path_ave, int16_t *table, int16_t *score)
{ static const double bias_factor[10] = { .690, .690, .690, .690, .780, .850, .900, .933, .966, 1.000 };
  double match = freq[0] + freq[3];
  int bias, ms;
  (void) tspace;
  if ((match <= 0.) == (match > 0.))
    match = .5;
  if (match > .5)
    match = 1.-match;
  bias = (int) ((match+.025)*20.-1.);
  if (match < .2)
    bias = 3;
  *path_ave = (int) (PATH_LEN * (1. - bias_factor[bias] * (1. - ave_corr)));
  ms = (int) (1000 * bias_factor[bias] * (1. - ave_corr));
  fill(0,0,0,0,ms,1000-ms,table,score);
  return 0;
}

void fga_alns_free(fga_alns *A)
{ if (A == NULL) return;
  free(A->alns); free(A->tbytes); free(A->ctg_waves); free(A);
}
