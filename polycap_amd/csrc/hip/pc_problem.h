/*
 * pc_problem.h -- host-side construction of the device tables from a pc_hip_problem (plain C++, no HIP).
 *
 * Derived tables (all fp64, nmax+1 entries):
 *   zh[i]   = ext[i] / (2 cos(pi/6) (n_shells+1))     hex-grid pitch: capillary axis = (kx,ky)*zh[i]
 *                                                     (reference src/polycap-photon.c:624-627)
 *   cap2[i] = cap[i]^2
 *   idz[i]  = 1 / (z[i+1] - z[i])
 *   hexd[i] = sqrt(ext^2 - (ext/2)^2)                 centre-to-edge distance of the outer hexagon
 *                                                     (reference src/polycap-photon.c:158)
 * Per-energy constants: complex refractive index n = (1-alfa) + i beta and (1/n)^2
 * (reference src/polycap-capil.c:497-503), roughness coefficient 1.01358*E*sig_rough (:626).
 */
#ifndef PC_PROBLEM_H
#define PC_PROBLEM_H

#include <cmath>
#include <string>
#include <vector>

#include "polycap-hip.h"
#include "pc_device.h"

#define PC_HC     1.23984193E-7
#define PC_N_AVOG 6.022098e+23
#define PC_R0     2.8179403227e-13

struct pc_host_tables {
	std::vector<double> z, cap, zh, cap2, hexd, idz, ext, stp, istp;
	std::vector<pc_drdev> dr;                 /* chord deviations of cap (leak path) */
	std::vector<float> mb1, md1, mb2, md2;   /* block-certificate tables for strides PC_L1, PC_L2 */
	std::vector<pc_marg4> mg;                /* the same, packed per node (what pc_march_ok reads) */
	std::vector<pc_energy_const> ec;
	std::vector<double> amu;                 /* linear attenuation coefficient per energy (leak path) */
	pc_params pm;
};

static inline int pc_build_tables(const pc_hip_problem *p, pc_host_tables &t, std::string &err)
{
	if (!p || !p->z || !p->cap || !p->ext) { err = "problem/profile arrays must not be NULL"; return PC_HIP_ERR_INVALID; }
	if (p->nmax < 1) { err = "nmax must be >= 1"; return PC_HIP_ERR_INVALID; }
	if (p->n_energies < 1 || !p->energies || !p->amu || !p->scatf) { err = "n_energies must be >= 1 with energies/amu/scatf tables"; return PC_HIP_ERR_INVALID; }
	if (p->n_cap < 1) { err = "n_cap must be >= 1"; return PC_HIP_ERR_INVALID; }
	const int n = p->nmax + 1;
	if (!(p->z[0] >= 0.)) { err = "z[0] must be >= 0"; return PC_HIP_ERR_INVALID; }
	for (int i = 0; i < n; i++) {
		if (i > 0 && !(p->z[i] > p->z[i-1])) { err = "profile z must be strictly increasing"; return PC_HIP_ERR_INVALID; }
		if (!(p->cap[i] >= 0.)) { err = "profile cap must be >= 0"; return PC_HIP_ERR_INVALID; }
	}
	pc_params &pm = t.pm;
	pm = pc_params();
	pm.nmax = p->nmax;
	pm.n_shells = std::round(std::sqrt(12. * (double)p->n_cap - 3.)/6. - 0.5);   /* src/polycap-photon.c:483 */
	pm.mono = (pm.n_shells == 0.) ? 1 : 0;
	pm.n_energies = (int)p->n_energies;
	pm.literal = 0;
	/* which Fresnel form a run evaluates depends on where its weights live (pc_launch_kernel): registers for 1, up to 4 and up to 8
	 * energies on profiles of up to 1024 points (FORMs 0 / 1), memory otherwise (FORM 3) */
	pm.form3 = (p->n_energies > 1 && !(p->n_energies <= 8 && p->nmax + 1 <= 1024)) ? 1 : 0;
	pm.hexscale = 2.*PC_COSPI_6*(pm.n_shells + 1);
	pm.inv_hexscale = 1.0/pm.hexscale;
	pm.uniform_illum = (p->src_sigx < 0. || p->src_sigy < 0.) ? 1 : 0;
	pm.generic_src = (p->src_x == p->src_y) ? 0 : 1;
	pm.d_source = p->d_source; pm.src_x = p->src_x; pm.src_y = p->src_y;
	pm.src_sigx = p->src_sigx; pm.src_sigy = p->src_sigy;
	pm.src_shiftx = p->src_shiftx; pm.src_shifty = p->src_shifty;
	pm.frac_hor_pol = (1. + p->hor_pol)/2.;                                       /* src/polycap-source.c:114 */
	pm.z_end = p->z[p->nmax];
	pm.ext_end = p->ext[p->nmax];
	pm.cap0 = p->cap[0];
	pm.ext0 = p->ext[0];

	t.z.assign(p->z, p->z + n);
	t.cap.assign(p->cap, p->cap + n);
	t.ext.assign(p->ext, p->ext + n);
	t.zh.resize(n); t.cap2.resize(n); t.hexd.resize(n); t.idz.assign(n, 0.); t.stp.resize(n); t.istp.resize(n);
	double dr2max = 0., capmin = HUGE_VAL, capmax = 0., extmax = 0., ratio = 0.;
	for (int i = 0; i < n; i++) {
		double e = p->ext[i], c = p->cap[i];
		t.zh[i] = e / pm.hexscale;
		t.cap2[i] = c*c;
		t.stp[i] = c/10.;
		t.istp[i] = 10./c;
		t.hexd[i] = (e > 0.) ? std::sqrt(e*e - (e/2.)*(e/2.)) : 0.;
		if (i + 1 < n) {
			t.idz[i] = 1.0 / (p->z[i+1] - p->z[i]);
			double dr = p->cap[i+1] - c;
			if (dr*dr > dr2max) dr2max = dr*dr;
		}
		if (c < capmin) capmin = c;
		if (c > capmax) capmax = c;
		if (std::fabs(e) > extmax) extmax = std::fabs(e);
		if (e > 0. && c/e > ratio) ratio = c/e;
		if (!(e > 0.)) ratio = HUGE_VAL;          /* degenerate exterior: keep every hexagon test literal */
	}
	double m = std::fmax(1e-6*capmin*capmin, 1e-10*capmax*extmax);
	pm.adj = 0.25*dr2max + m;
	pm.two_rmax = 2.*capmax;
	/* float copies for the comparisons of pc_march_ok: rounded up, then inflated */
	pm.adjf = std::nextafter((float)pm.adj, HUGE_VALF) * PC_MARGIN_INFLATE;
	pm.two_rmaxf = std::nextafter((float)pm.two_rmax, HUGE_VALF) * PC_MARGIN_INFLATE;
	/* block certificates: for every start node the chord deviations over the next L segments */
	for (int lvl = 1; lvl <= 2; lvl++) {
		const int L = (lvl == 1) ? PC_L1 : PC_L2;
		std::vector<float> &mb = (lvl == 1) ? t.mb1 : t.mb2;
		std::vector<float> &md = (lvl == 1) ? t.md1 : t.md2;
		mb.assign(n, HUGE_VALF);
		md.assign(n, HUGE_VALF);
		if (lvl == 1) t.dr.assign(n, pc_drdev{HUGE_VALF, HUGE_VALF});
		for (int i = 0; i + L < n; i++) {
			const double za = p->z[i], zb = p->z[i+L], span = zb - za;
			double dzh = 0., dr = 0.;
			for (int j = i + 1; j < i + L; j++) {
				double u = (p->z[j] - za)/span;
				double zc = t.zh[i] + (t.zh[i+L] - t.zh[i])*u;
				double rc = p->cap[i] + (p->cap[i+L] - p->cap[i])*u;
				dzh = std::fmax(dzh, std::fabs(t.zh[j] - zc));
				dr = std::fmax(dr, std::fabs(p->cap[j] - rc));
			}
			/* 1e-12 relative slack covers the rounding of the chord evaluation itself */
			dzh = dzh*(1. + 1e-9) + 1e-12*std::fabs(t.zh[i]);
			dr = dr*(1. + 1e-9) + 1e-12*capmax;
			double dR = p->cap[i+L] - p->cap[i];
			double rblk = 0.;
			for (int j = i; j <= i + L; j++) rblk = std::fmax(rblk, p->cap[j]);
			double base = 0.25*dR*dR + 2.*rblk*dr + m;
			float fb = (float)base, fd = (float)dzh;
			if ((double)fb < base) fb = std::nextafter(fb, HUGE_VALF);
			if ((double)fd < dzh) fd = std::nextafter(fd, HUGE_VALF);
			mb[i] = fb;
			md[i] = fd;
			float fr = (float)dr;
			if ((double)fr < dr) fr = std::nextafter(fr, HUGE_VALF);
			if (lvl == 1) t.dr[i].d1 = fr; else t.dr[i].d2 = fr;
		}
	}
	/* packed per start node (pc_marg4): margin bases inflated, rounded up and cut to their upper 16 bits (rounded up again:
	 * positive floats order like their bit patterns; infinity stays infinity), twice the largest radius of the wider block */
	t.mg.resize(n);
	for (int i = 0; i < n; i++) {
		unsigned int half[2];
		for (int lvl = 0; lvl < 2; lvl++) {
			const float v = std::nextafter(((lvl == 0) ? t.mb1[i] : t.mb2[i]) * PC_MARGIN_INFLATE, HUGE_VALF);
			unsigned int u = __builtin_bit_cast(unsigned int, v);
			if (u & 0xffffu) u = (u & 0xffff0000u) + 0x10000u;
			half[lvl] = u >> 16;
		}
		double rblk = 0.;
		for (int j = i; j < n && j <= i + PC_L2; j++) rblk = std::fmax(rblk, p->cap[j]);
		const float r2 = std::nextafter((float)(2.*rblk), HUGE_VALF) * PC_MARGIN_INFLATE;
		t.mg[i] = pc_marg4{(half[0] << 16) | half[1], t.md1[i], t.md2[i], std::nextafter(r2, HUGE_VALF)};
	}
	pm.bnd_thresh = ratio + 1e-9;

	t.ec.resize(p->n_energies);
	t.amu.clear();
	for (size_t k = 0; k < p->n_energies; k++) {
		double e = p->energies[k], scatf = p->scatf[k], amu = p->amu[k];
		t.amu.push_back(amu);
		pc_energy_const &c = t.ec[k];
		double alfa = (PC_HC/e)*(PC_HC/e)*((PC_N_AVOG*PC_R0*p->density)/(2*PC_PI)) * scatf;
		double beta = (PC_HC)/(4.*PC_PI) * (amu/e);
		/* n = (1 - alfa) + i beta, 1/n and (1/n)^2 in plain real arithmetic: std::complex division and multiplication are
		 * implemented differently by different compilers/runtimes (scaled library calls or inline formulas), and these
		 * constants must come out the same wherever this header is compiled */
		const double nre = 1.0 - alfa, nim = beta;
		const double nn2 = nre*nre + nim*nim;
		const double ire = nre/nn2, iim = -nim/nn2;
		c.n_re = nre; c.n_im = nim;
		c.ninv2_re = ire*ire - iim*iim; c.ninv2_im = 2.*(ire*iim);
		c.rough_c = (1.01358e0*e)*p->sig_rough;
		/* FORM 3 (pc_fresnel3): n^2 and d2 = 1 - Re n^2 = alfa (2 - alfa) + beta^2 */
		c.d2 = alfa*(2.0 - alfa) + beta*beta;
		c.n2_re = 1.0 - c.d2;
		c.n2_im = 2.*(nre*nim);
		c.zi2 = std::fmax(c.n2_im*c.n2_im, 6.223015277861142e-61 /* 2^-200 */);
		c.rough_k2 = c.rough_c*c.rough_c;
		/* argument checks of polycap_refl_polar, src/polycap-capil.c:463-478 */
		c.valid = (e >= 1. && e <= 100. && p->density > 0. && scatf >= 0. && amu >= 0.) ? 1. : 0.;
	}
	return PC_HIP_OK;
}

#endif
