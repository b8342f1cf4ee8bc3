// facade_demo.cpp — the reference application's call sequence (main.cpp:211-218,262,273,286) through the C++ facade
// (optixpathtracer_amd/csrc/SampleRenderer.h) over the C ABI: no Python, no torch, no hipcc on the application side.
//   g++ -std=c++17 -I<repo> -I<repo>/include facade_demo.cpp -L<repo>/optixpathtracer_amd -lptamd -o facade_demo
//   ./facade_demo scene.bin out.bin [ncontexts]      ncontexts > 0: MultiSampleRenderer with that many contexts on device 0
// scene.bin (little endian): u32 nmesh; per mesh { u32 nv, u32 nt, Material (104 B), nv*3 f32, nt*3 u32 };
//   u32 probe_w, probe_h, probe_w*probe_h*4 f32; f32 eye[3], lookat[3], up[3], fovY; u32 width, height, spp, subframes
// out.bin: width*height u32 rgba8 frame, then width*height*4 f32 accum_buffer.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "optixpathtracer_amd/csrc/SampleRenderer.h"

using namespace ptamd;

template <typename T>
static void rd(FILE* f, T* p, size_t n) {
    if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s scene.bin out.bin [ncontexts [frames_in_flight]]\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    try {
        Model model;
        uint32_t nmesh;
        rd(f, &nmesh, 1);
        for (uint32_t m = 0; m < nmesh; ++m) {
            TriangleMesh* mesh = new TriangleMesh;
            uint32_t nv, nt;
            rd(f, &nv, 1);
            rd(f, &nt, 1);
            rd(f, &mesh->material, 1);
            mesh->vertex.resize(nv);
            mesh->index.resize(nt);
            rd(f, mesh->vertex.data(), nv);
            rd(f, mesh->index.data(), nt);
            model.meshes.push_back(mesh);
        }
        uint32_t pw, ph;
        rd(f, &pw, 1);
        rd(f, &ph, 1);
        std::vector<Color> pdata((size_t)pw * ph);
        rd(f, pdata.data(), pdata.size());
        Camera cam;
        rd(f, &cam.eye, 1);
        rd(f, &cam.lookat, 1);
        rd(f, &cam.up, 1);
        rd(f, &cam.fovY, 1);
        uint32_t w, h, spp, subframes;
        rd(f, &w, 1); rd(f, &h, 1); rd(f, &spp, 1); rd(f, &subframes, 1);
        fclose(f);

        ProbeData probe;                                // main.cpp:146-156 (loadProbe) + BuildCDF
        probe.width = (int)pw; probe.height = (int)ph; probe.data = pdata.data();
        probe.BuildCDF();
        const int ncontexts = argc > 3 ? atoi(argv[3]) : 0;
        if (ncontexts > 0) {                            // the same application code on several contexts of one process
            MultiSampleRenderer multi(&model, std::vector<int>((size_t)ncontexts, 0));
            multi.setProbe(probe);
            multi.resize(int2{(int)w, (int)h});
            cam.aspectRatio = (float)w / (float)h;
            multi.setCamera(cam);
            multi.launchParams.samples_per_launch = spp;
            std::vector<uint32_t> pixels((size_t)w * h);
            const int mfif = argc > 4 ? atoi(argv[4]) : 0;  // 2 or 3: renderToHost(pixels) shows frame k-1 while frame k renders (overlapped hand-over)
            if (mfif >= 2) multi.setFramesInFlight(mfif);
            for (uint32_t s = 0; s < subframes; ++s) {
                multi.launchParams.frame.subframe_index = s;
                multi.renderToHost(pixels.data());
            }
            if (mfif >= 2) multi.flush(pixels.data());      // the last frame goes on display
            multi.gather(PT_BUF_ACCUM);
            std::vector<float> accum((size_t)w * h * 4);
            if (pt_download(pt_multi_ctx(multi.multi, ncontexts - 1), PT_BUF_ACCUM, accum.data(), accum.size() * sizeof(float)) != PT_OK) throw std::runtime_error("download");
            FILE* o = fopen(argv[2], "wb");
            if (!o) { perror(argv[2]); return 2; }
            fwrite(pixels.data(), sizeof(uint32_t), pixels.size(), o);
            fwrite(accum.data(), sizeof(float), accum.size(), o);
            fclose(o);
            pt_multi_stats ms;
            pt_multi_get_stats(multi.multi, &ms);
            printf("%d contexts: %ux%u, %u spp x %u subframes: last frame %.3f ms (max over ranks), gather %.3f ms (exchange kind %d), %llu rays\n", ncontexts, w, h, spp,
                   subframes, ms.sum.render_ms, ms.gather_ms, ms.exchange, (unsigned long long)(ms.sum.radiance_rays + ms.sum.shadow_rays));
            return 0;
        }
        SampleRenderer sample(&model);                  // main.cpp:211
        sample.setProbe(probe);                         // main.cpp:216
        sample.resize(int2{(int)w, (int)h});            // main.cpp:218
        cam.aspectRatio = (float)w / (float)h;
        sample.setCamera(cam);                          // main.cpp:262
        sample.launchParams.samples_per_launch = spp;
        std::vector<uint32_t> pixels((size_t)w * h);
        const int fif = argc > 4 ? atoi(argv[4]) : 0;      // 2 or 3 frames in flight: render() returns while its frame is still running
        const bool pipelined = fif >= 2;
        if (pipelined) sample.setFramesInFlight(fif);
        for (uint32_t s = 0; s < subframes; ++s) {      // the render loop, main.cpp:273-286
            sample.launchParams.frame.subframe_index = s;
            if (pipelined) sample.render();
            else sample.renderToHost(pixels.data());
        }
        if (pipelined) sample.downloadPixels(pixels.data()); // waits for the last frame
        std::vector<float> accum((size_t)w * h * 4);
        if (pt_download(sample.ctx, PT_BUF_ACCUM, accum.data(), accum.size() * sizeof(float)) != PT_OK) throw std::runtime_error(pt_last_error(sample.ctx));
        FILE* o = fopen(argv[2], "wb");
        if (!o) { perror(argv[2]); return 2; }
        fwrite(pixels.data(), sizeof(uint32_t), pixels.size(), o);
        fwrite(accum.data(), sizeof(float), accum.size(), o);
        fclose(o);
        pt_stats st;
        pt_get_stats(sample.ctx, &st);
        printf("%ux%u, %u spp x %u subframes: last frame %.3f ms, %llu rays\n", w, h, spp, subframes, st.render_ms,
               (unsigned long long)(st.radiance_rays + st.shadow_rays));
    } catch (const std::exception& e) {                 // main.cpp:314-317
        fprintf(stderr, "Caught exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
