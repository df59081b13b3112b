/* tools/verify/asan_planner.c -- the host-side planners under AddressSanitizer / UBSan (CPU build only; the GPU pool has no
 * sanitizers). Reads every .ednn blob named on the command line and runs what a model load runs on the host: ed_plan_net,
 * ed_plan_net_mm (through edison_net_plan_dump) and the specialisation header (edison_net_spec_source); then feeds the loaders
 * truncated copies of the blob, which they must refuse without reading past the end.
 *   tools/verify/asan_planner.sh   builds this with the library's C sources and feeds it the fixture graphs + random ones */
#include <stdio.h>
#include <stdlib.h>

#include "../../include/edison_hip.h"
#include "../../edison_amd/csrc/edison_internal.h"

/* the table builders and the kws_conv model parser: every filterbank the tests configure, every frame length of variant D */
static int host_tables(const unsigned char *shipped, long shipped_bytes)
{
	char err[256];
	int built = 0;
	static const double edges[][4] = {{16000, 80, 7600, 128}, {16000, 0, 8000, 128}, {16000, 300, 3400, 64}, {8000, 20, 4000, 100}, {44100, 80, 7600, 128}, {16000, 7000, 7600, 128}};
	ed_mfcc_tables_t *t = (ed_mfcc_tables_t *)malloc(sizeof(*t));
	ed_q15_tables_t *q = (ed_q15_tables_t *)malloc(sizeof(*q));
	ed_f32_tables_t *f = (ed_f32_tables_t *)malloc(sizeof(*f));
	for (size_t e = 0; e < sizeof(edges) / sizeof(edges[0]); e++)
	{
		for (int v = 0; v < 2; v++) built += ed_build_mfcc_tables(v, edges[e][0], edges[e][1], edges[e][2], edges[e][3], t, err, sizeof(err)) == 0;
		built += ed_build_q15_tables(edges[e][0], edges[e][1], edges[e][2], edges[e][3], q, err, sizeof(err)) == 0; /* may refuse: that is an answer too */
		double *W = (double *)malloc(sizeof(double) * 513 * 32);
		(void)ed_gen_mel_weight_matrix(32, 513, edges[e][0], edges[e][1], edges[e][2], W);
		free(W);
	}
	static const int lens[] = {2, 3, 64, 255, 256, 400, 480, 512, 640, 1000, 1024, 2048, 4096, 5000};
	for (size_t k = 0; k < sizeof(lens) / sizeof(lens[0]); k++)
		for (int nf = 1; nf <= 26; nf += 6)
			built += ed_build_f32_tables(nf, nf > 3 ? 1 : 0, lens[k], 8, 0.97f, f, err, sizeof(err)) == 0;
	ed_cnn_model_t *m = (ed_cnn_model_t *)malloc(sizeof(*m));
	ed_cnn_mfma_model_t *mm = (ed_cnn_mfma_model_t *)malloc(sizeof(*mm));
	if (ed_parse_model(shipped, (size_t)shipped_bytes, m, mm, err, sizeof(err)) != 0) { fprintf(stderr, "shipped model refused: %s\n", err); return -1; }
	for (long cut = 0; cut < shipped_bytes; cut += (cut < 600 ? 5 : 1 + shipped_bytes / 60))
	{
		unsigned char *part = (unsigned char *)malloc((size_t)cut + 1);
		for (long k = 0; k < cut; k++) part[k] = shipped[k];
		if (ed_parse_model(part, (size_t)cut, m, mm, err, sizeof(err)) == 0) { fprintf(stderr, "a truncated model (%ld bytes) was accepted\n", cut); return -1; }
		free(part);
	}
	free(t); free(q); free(f); free(m); free(mm);
	return built;
}

int main(int argc, char **argv)
{
	int planned = 0, refused = 0, tables = -1;
	for (int i = 1; i < argc; i++)
	{
		FILE *f = fopen(argv[i], "rb");
		if (!f) { fprintf(stderr, "cannot open %s\n", argv[i]); return 2; }
		fseek(f, 0, SEEK_END);
		const long n = ftell(f);
		fseek(f, 0, SEEK_SET);
		unsigned char *blob = (unsigned char *)malloc((size_t)n);
		if (!blob || fread(blob, 1, (size_t)n, f) != (size_t)n) return 2;
		fclose(f);
		size_t fneed = 0, sneed = 0, tneed = 0;
		const size_t psz = edison_net_plan_layout(0), msz = edison_net_plan_layout(1);
		void *plan = malloc(psz), *mm = malloc(msz);
		int r = edison_net_plan_dump(blob, (size_t)n, plan, psz, mm, msz, NULL, 0, &fneed, NULL, 0, &sneed);
		if (r == EDISON_OK)
		{
			void *frag = malloc(fneed + 1), *seeds = malloc(sneed + 4);
			r = edison_net_plan_dump(blob, (size_t)n, NULL, 0, NULL, 0, frag, fneed, &fneed, seeds, sneed, &sneed);
			if (r != EDISON_OK) { fprintf(stderr, "%s: second dump failed %d\n", argv[i], r); return 1; }
			if (edison_net_spec_source(blob, (size_t)n, NULL, 0, &tneed) != EDISON_E_SIZE) return 1;
			char *text = (char *)malloc(tneed);
			if (edison_net_spec_source(blob, (size_t)n, text, tneed, &tneed) != EDISON_OK) return 1;
			free(text); free(frag); free(seeds);
			planned++;
		}
		else refused++;
		if (tables < 0 && i == 1) /* the first blob is the shipped model (asan_planner.sh) */
		{
			tables = host_tables(blob, n);
			if (tables < 0) return 1;
		}
		for (long cut = 0; cut < n; cut += (cut < 400 ? 7 : 1 + n / 40))
		{
			unsigned char *part = (unsigned char *)malloc((size_t)cut + 1); /* an exact-size copy: an overrun is a heap overflow */
			for (long k = 0; k < cut; k++) part[k] = blob[k];
			(void)edison_net_plan_dump(part, (size_t)cut, NULL, 0, NULL, 0, NULL, 0, NULL, NULL, 0, NULL);
			free(part);
		}
		free(plan); free(mm); free(blob);
	}
	printf("asan_planner: %d graphs planned, %d refused, truncated copies refused, %d table sets built, no sanitizer report\n", planned, refused, tables);
	return 0;
}
