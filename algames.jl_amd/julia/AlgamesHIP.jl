# AlgamesHIP.jl -- Julia host shim over the C ABI of libalgames_hip.so (include/algames_hip.h).
#
# UNTESTED IN THIS ENVIRONMENT: neither the build container nor the GPU box has a `julia` binary (SURVEY.md section 0), so
# this file is the binding a maintainer of Algames.jl would add.  Every ABI call it makes, in the order it makes them, is
# mirrored through the identical ctypes binding by tests/test_gpu_boundary.py (same layouts, same write-back arithmetic).
#
# It keeps the reference API surface for the hot path.  `GameProblem` / `Options` are the reference's own types;
#   bp = BatchedGameProblem(probs; device=0)          one device handle for a batch of structurally identical problems
#   newton_solve!(bp)                                 src/problem/solver_methods.jl:5-65 for every problem of the batch
#   ibr_newton_solve!(bp; ibr_opts), ibr_newton_solve!(bp, i)     solver_methods.jl:133-228
#   mpc_solve!(bp, steps)                             receding-horizon loop of BASELINE config 5 (opts.shift / opts.dual_reset)
#   newton_solve!(probs::Vector{<:GameProblem})       convenience: build a handle, solve, release
#   sp = ShardedGameProblem(probs; devices=0:7); newton_solve!(sp)     the batch split over several devices (one handle each)
# After a solve every `prob.pdtraj`, the multipliers / penalties / values of every constraint (`conval.λ`, `.μ`, `.vals`,
# constraints_methods.jl:329-379) and `prob.stats` (struct/statistics.jl:5-57) hold what the reference's solver would have left.
module AlgamesHIP

using Algames
using LinearAlgebra
using StaticArrays
import Algames: newton_solve!, ibr_newton_solve!

const TO = Algames.TrajectoryOptimization
const LIB = get(ENV, "ALGAMES_HIP_LIB", joinpath(@__DIR__, "..", "lib", "libalgames_hip.so"))

struct AlgDesc
    model::Int32; p::Int32; d::Int32; N::Int32
    dt::Float64
    batch::Int32; device::Int32
end

struct AlgOptions                      # POD mirror of include/algames_hip.h: alg_options
    amplitude_init::Float64; shift::Int32; regularize::Int32
    reg_0::Float64; alpha_decrease::Float64; beta::Float64
    ls_iter::Int32; dual_reset::Int32; delta_min::Float64
    rho_0::Float64; rho_increase::Float64; rho_max::Float64; lambda_max::Float64; alpha_dual::Float64
    alphax_dual::NTuple{10,Float64}
    eps_dyn::Float64; eps_sta::Float64; eps_con::Float64; eps_opt::Float64
    outer_iter::Int32; inner_iter::Int32
    seed::Int64
end

struct AlgRecord
    outer::Int32; ls_j::Int32
    alpha::Float64; res::Float64; delta::Float64
    dyn_vio::Float64; con_vio::Float64; sta_vio::Float64; opt_vio::Float64
    t_elap::Float64
end

struct AlgGameStats
    status::Int32; outer_iters::Int32; newton_iters::Int32; records::Int32; converged::Int32; ls_failures::Int32
    refinements::Int32; reserved::Int32
    last::AlgRecord
end

check(rc) = rc == 0 || error(unsafe_string(ccall((:alg_last_error, LIB), Cstring, ())))

model_id(::DoubleIntegratorGame) = Int32(0)
model_id(::UnicycleGame) = Int32(1)
model_id(::BicycleGame) = Int32(2)
model_id(::QuadrotorGame) = Int32(3)      # src/dynamics/quadrotor.jl (the constructor's constants; mass through alg_set_quadrotor)

function abi_options(o::Options)
    ax = ntuple(i -> i <= length(o.αx_dual) ? Float64(o.αx_dual[i]) : 1.0, 10)
    AlgOptions(o.amplitude_init, min(o.shift, 2^30), o.regularize, o.reg_0, o.α_decrease, o.β, o.ls_iter,
               o.dual_reset, o.Δ_min, o.ρ_0, o.ρ_increase, o.ρ_max, o.λ_max, o.α_dual, ax,
               o.ϵ_dyn, o.ϵ_sta, o.ϵ_con, o.ϵ_opt, o.outer_iter, o.inner_iter, o.seed)
end

"""
A batch of structurally identical `GameProblem`s bound to one device handle (`alg_create` ... `alg_destroy`).
All problems must share model, N, dt, options and constraint structure (collision-avoidance radii, control / state bounds,
walls, circles, collision cost); they may differ in x0 and in the LQR data.  The handle lives until `close(bp)` or finalisation,
so repeated solves (warm starts with `opts.dual_reset = false`, MPC loops) reuse the device buffers.
"""
mutable struct BatchedGameProblem{P<:GameProblem}
    probs::Vector{P}
    h::Ptr{Cvoid}
    con_len::Int
    function BatchedGameProblem(probs::Vector{P}; device::Integer=0) where {P<:GameProblem}
        bp = new{P}(probs, C_NULL, 0)
        finalizer(close, bp)                                        # registered first: a failing setup! must not leak the handle
        try
            setup!(bp, device)
        catch
            close(bp)
            rethrow()
        end
        return bp
    end
end

function Base.close(bp::BatchedGameProblem)
    if bp.h != C_NULL
        ccall((:alg_destroy, LIB), Cvoid, (Ptr{Cvoid},), bp.h)
        bp.h = C_NULL
    end
    return nothing
end

sync_options!(bp::BatchedGameProblem) =       # `opts` is shared by reference and read at solve time, like the reference does
    check(ccall((:alg_set_options, LIB), Cint, (Ptr{Cvoid}, Ref{AlgOptions}), bp.h, Ref(abi_options(bp.probs[1].opts))))

function setup!(bp::BatchedGameProblem, device)
    probs = bp.probs; prob = probs[1]; ps = prob.probsize; B = length(probs)
    N, n, m, p = ps.N, ps.n, ps.m, ps.p
    ni, mi = ps.ni[1], ps.mi[1]
    d = prob.model isa DoubleIntegratorGame ? mi : (prob.model isa QuadrotorGame ? 3 : 2)
    dt = prob.pdtraj.pr[1].dt
    desc = Ref(AlgDesc(model_id(prob.model), p, d, N, dt, B, device))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:alg_create, LIB), Cint, (Ref{AlgDesc}, Ref{Ptr{Cvoid}}), desc, h))
    bp.h = h[]
    sync_options!(bp)
    # x0: B x n, game-major (Julia is column-major: build n x B)
    x0 = hcat([Vector(pr.x0) for pr in probs]...)
    check(ccall((:alg_set_x0, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), bp.h, x0))
    # LQR data from game_obj.obj[i][1] (LQRCost: Q diagonal, q = -Q xf), per game: (ni x p x B) in memory order
    Qd = zeros(ni, p, B); Rd = zeros(mi, p, B); xf = zeros(ni, p, B); uf = zeros(mi, p, B)
    for (g, pr) in enumerate(probs), i in 1:p
        c = pr.game_obj.obj[i][1].cost[1]
        Qfull = diag(c.Q); Rfull = diag(c.R)
        Qd[:, i, g] = Qfull[ps.pz[i]]; Rd[:, i, g] = Rfull[ps.pu[i]]
        xf[:, i, g] = -(c.q ./ map(x -> x == 0 ? 1.0 : x, Qfull))[ps.pz[i]]
        uf[:, i, g] = -(c.r ./ map(x -> x == 0 ? 1.0 : x, Rfull))[ps.pu[i]]
    end
    check(ccall((:alg_set_lqr, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32),
                bp.h, Qd, Rd, xf, uf, 1))
    # collision costs: game_obj.obj[i][2:end] are CollisionCost objectives (objective.jl:84-100)
    if length(prob.game_obj.obj[1]) > 1
        rad = [prob.game_obj.obj[i][2].cost[1].r for i in 1:p]
        mu = [prob.game_obj.obj[i][2].cost[1].μ for i in 1:p]
        check(ccall((:alg_add_collision_cost, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), bp.h, rad, mu))
    end
    # state constraints of player i, in the order they were added (constraints_methods.jl): dispatch on the type
    # collision avoidance: every CollisionConstraint of player i's state constraints is ONE ordered pair (i, j) with its own radius
    # (add_collision_avoidance!(game_con, i, j, radius), constraints_methods.jl:5-19; the vector form adds all pairs with r_i + r_j,
    # :21-33).  add_spherical_collision_avoidance! builds the constraint on pz[i][1:3] (three indices) instead of px[i] (two).
    for i in 1:p, cv in prob.game_con.state_conval[i]
        con = cv.con
        con isa TO.CollisionConstraint || continue
        j = partner_of(ps, con)
        fn = length(con.x1) == 3 ? :alg_add_spherical_collision_avoidance_pair : :alg_add_collision_avoidance_pair
        check(fn === :alg_add_collision_avoidance_pair ?
              ccall((:alg_add_collision_avoidance_pair, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Float64), bp.h, i - 1, j - 1, con.radius) :
              ccall((:alg_add_spherical_collision_avoidance_pair, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Float64), bp.h, i - 1, j - 1, con.radius))
    end
    if prob.model isa BicycleGame
        check(ccall((:alg_set_bicycle, LIB), Cint, (Ptr{Cvoid}, Float64, Float64), bp.h, prob.model.lf, prob.model.lr))
    elseif prob.model isa QuadrotorGame
        check(ccall((:alg_set_quadrotor, LIB), Cint, (Ptr{Cvoid}, Float64), bp.h, prob.model.mass))
    end
    pts(a, b, c) = Vector{Float64}(vec(permutedims(hcat(Vector(a), Vector(b), Vector(c)))))      # n x 3, row-major
    for i in 1:p, cv in prob.game_con.state_conval[i]
        con = cv.con
        if con isa Algames.StateBoundConstraint                      # add_state_bound!(game_con, i, x_max, x_min)
            check(ccall((:alg_add_state_bound, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}), bp.h, i - 1,
                        Vector{Float64}(con.x_max), Vector{Float64}(con.x_min)))
        elseif con isa Algames.WallConstraint                        # add_wall_constraint!(game_con, walls) / (game_con, i, walls)
            check(ccall((:alg_add_wall_constraint_player, LIB), Cint,
                        (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), bp.h, i - 1, length(con),
                        Vector{Float64}(con.x1), Vector{Float64}(con.y1), Vector{Float64}(con.x2), Vector{Float64}(con.y2),
                        Vector{Float64}(con.xv), Vector{Float64}(con.yv)))
        elseif con isa TO.CircleConstraint                           # add_circle_constraint!(game_con, xc, yc, radius) / (game_con, i, ...)
            check(ccall((:alg_add_circle_constraint_player, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), bp.h, i - 1, length(con),
                        Vector{Float64}(con.x), Vector{Float64}(con.y), Vector{Float64}(con.radius)))
        elseif con isa Algames.Wall3DConstraint                      # add_wall_constraint!(game_con, walls::Vector{Wall3D}) / (game_con, i, walls) (constraints_methods.jl:208-247)
            check(ccall((:alg_add_wall3d_constraint_player, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), bp.h, i - 1, length(con),
                        pts(con.x1, con.y1, con.z1), pts(con.x2, con.y2, con.z2), pts(con.x3, con.y3, con.z3), pts(con.xv, con.yv, con.zv)))
        elseif con isa Algames.CylinderConstraint                    # add_wall_constraint!(game_con, walls::Vector{CylinderWall}) / (game_con, i, walls) (:256-299)
            axis = Int32[s == :x ? 0 : s == :y ? 1 : 2 for s in con.v]
            check(ccall((:alg_add_cylinder_constraint_player, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}), bp.h, i - 1, length(con),
                        pts(con.p1, con.p2, con.p3), axis, Vector{Float64}(con.l), Vector{Float64}(con.r)))
        end
    end
    if !isempty(prob.game_con.control_conval)
        con = prob.game_con.control_conval[1].con
        check(ccall((:alg_add_control_bound, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), bp.h,
                    Vector(con.u_max), Vector(con.u_min)))
    end
    cl = Ref{Int32}(0)
    check(ccall((:alg_get_con_len, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}), bp.h, cl))
    bp.con_len = cl[]
    return bp
end

# player j whose position indices a CollisionConstraint of player i points at (x2 = px[j] or pz[j][1:3])
partner_of(ps, con) = findfirst(q -> all(con.x2 .== ps.px[q][1:length(con.x2)]) || all(con.x2 .== ps.pz[q][1:length(con.x2)]), 1:ps.p)

# The library keeps ONE table of distinct walls (circles) per handle; alg_add_wall_constraint_player appends the entries it has not
# seen, in call order, and row w of the ABI's wall block is table entry w for every player (include/algames_hip.h).  The same
# table is rebuilt here from the constraint objects, in the order setup! made the calls (players 1..p, their convals in order), so
# that a conval's row r maps to its table index.
wall_key(con, r) = (con.x1[r], con.y1[r], con.x2[r], con.y2[r], con.xv[r], con.yv[r])
circ_key(con, r) = (con.x[r], con.y[r], con.radius[r])
wall3_key(con, r) = (con.x1[r], con.y1[r], con.z1[r], con.x2[r], con.y2[r], con.z2[r], con.x3[r], con.y3[r], con.z3[r], con.xv[r], con.yv[r], con.zv[r])
cyl_key(con, r) = (con.p1[r], con.p2[r], con.p3[r], con.v[r], con.l[r], con.r[r])
function constraint_tables(prob)
    walls = Tuple[]; circs = Tuple[]; walls3 = Tuple[]; cyls = Tuple[]
    for i in 1:prob.probsize.p, cv in prob.game_con.state_conval[i]
        con = cv.con
        if con isa Algames.WallConstraint
            for r in 1:length(con); k = wall_key(con, r); k in walls || push!(walls, k); end
        elseif con isa TO.CircleConstraint
            for r in 1:length(con); k = circ_key(con, r); k in circs || push!(circs, k); end
        elseif con isa Algames.Wall3DConstraint
            for r in 1:length(con); k = wall3_key(con, r); k in walls3 || push!(walls3, k); end
        elseif con isa Algames.CylinderConstraint
            for r in 1:length(con); k = cyl_key(con, r); k in cyls || push!(cyls, k); end
        end
    end
    return walls, circs, walls3, cyls
end

# ---- layout of the ABI's constraint vectors (include/algames_hip.h "Layouts") -----------------------------------------------
# Returns, for constraint value object `cv` of player i (0 = shared control constraint), a function (l, r) -> 1-based position in
# the ABI vector of row r (1-based, in the conval's own row numbering) at the l-th knot index of the conval.
function abi_position(bp::BatchedGameProblem, cv, i::Int)
    prob = bp.probs[1]; ps = prob.probsize; N, n, m, p = ps.N, ps.n, ps.m, ps.p
    K = N - 1
    con = cv.con
    col_len = p * (p - 1) * K                                        # the ABI always carries the collision-avoidance and control-bound rows
    ctl_len = 2m * K                                                # (include/algames_hip.h "Layouts"); rows that were never added are inert
    has_sb = any(c -> c.con isa Algames.StateBoundConstraint, vcat(prob.game_con.state_conval...))
    sb_len = has_sb ? p * 2n * K : 0
    walls, circs, walls3, cyls = constraint_tables(prob)             # distinct entries in call order = the library's tables
    nwall, ncirc = length(walls), length(circs)
    if con isa TO.CollisionConstraint                               # rows: pair q = (i, j), knot k = 2..N
        j = partner_of(ps, con)
        q = (i - 1) * (p - 1) + (j < i ? j : j - 1) - 1
        return (l, r) -> q * K + (cv.inds[l] - 2) + 1
    elseif con isa Algames.ControlBoundConstraint                   # knot k = 1..N-1: (u - u_max)(m) then (u_min - u)(m); the reference keeps finite rows only
        fin = con.inds                                              # control_bound_constraint.jl:35-38
        return (l, r) -> col_len + (cv.inds[l] - 1) * 2m + fin[r]
    elseif con isa Algames.StateBoundConstraint                     # player i, knot k = 2..N: (x - x_max)(n) then (x_min - x)(n), finite rows
        fin = con.inds
        return (l, r) -> col_len + ctl_len + ((i - 1) * K + (cv.inds[l] - 2)) * 2n + fin[r]
    elseif con isa Algames.WallConstraint                           # row r of the conval = table entry w(r)
        tw = [findfirst(==(wall_key(con, r)), walls) for r in 1:length(con)]
        return (l, r) -> col_len + ctl_len + sb_len + ((i - 1) * K + (cv.inds[l] - 2)) * nwall + tw[r]
    elseif con isa TO.CircleConstraint
        tc = [findfirst(==(circ_key(con, r)), circs) for r in 1:length(con)]
        return (l, r) -> col_len + ctl_len + sb_len + p * nwall * K + ((i - 1) * K + (cv.inds[l] - 2)) * ncirc + tc[r]
    elseif con isa Algames.Wall3DConstraint || con isa Algames.CylinderConstraint
        nw3, ncy = length(walls3), length(cyls)                      # row r of the conval = table entry t(r), as for the planar walls
        base = col_len + ctl_len + sb_len + p * nwall * K + p * ncirc * K
        if con isa Algames.Wall3DConstraint
            t3 = [findfirst(==(wall3_key(con, r)), walls3) for r in 1:length(con)]
            return (l, r) -> base + ((i - 1) * K + (cv.inds[l] - 2)) * nw3 + t3[r]
        end
        ty = [findfirst(==(cyl_key(con, r)), cyls) for r in 1:length(con)]
        return (l, r) -> base + p * nw3 * K + ((i - 1) * K + (cv.inds[l] - 2)) * ncy + ty[r]
    end
    error("AlgamesHIP: constraint type $(typeof(con)) is not bound")
end

each_conval(f, game_con) = begin
    for (i, list) in enumerate(game_con.state_conval), cv in list; f(cv, i); end
    for cv in game_con.control_conval; f(cv, 0); end
end

"Push the multipliers / penalties the Julia side holds to the device (warm starts with opts.dual_reset = false)."
function push_duals!(bp::BatchedGameProblem)
    bp.con_len == 0 && return
    B = length(bp.probs)
    lam = zeros(bp.con_len, B); mu = fill(Float64(bp.probs[1].opts.ρ_0), bp.con_len, B)
    for (g, pr) in enumerate(bp.probs)
        each_conval(pr.game_con) do cv, i
            pos = abi_position(bp, cv, i)
            for l in eachindex(cv.inds), r in eachindex(cv.λ[l])
                lam[pos(l, r), g] = cv.λ[l][r]; mu[pos(l, r), g] = cv.μ[l][r]
            end
        end
    end
    check(ccall((:alg_set_con_duals, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), bp.h, lam, mu))
end

"Write back what a solve left on the device: pdtraj, constraint values / multipliers / penalties, Statistics."
function pull_results!(bp::BatchedGameProblem, stats::Vector{AlgGameStats})
    prob = bp.probs[1]; ps = prob.probsize; B = length(bp.probs)
    N, n, S = ps.N, ps.n, ps.S
    # pdtraj: [x_1 | horizontal-order vector] per game (primal_dual_traj.jl:46-75)
    z = zeros(n + S, B)
    check(ccall((:alg_get_traj, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), bp.h, 0, z))
    for (g, pr) in enumerate(bp.probs)
        Algames.set_traj!(pr.core, pr.pdtraj, view(z, n+1:n+S, g))
        Algames.RobotDynamics.set_state!(pr.pdtraj.pr[1], SVector{n}(z[1:n, g]))
    end
    # constraint multipliers and penalties (dual_update!, penalty_update!, constraints_methods.jl:329-379, 421-440)
    if bp.con_len > 0
        lam = zeros(bp.con_len, B); mu = zeros(bp.con_len, B)
        check(ccall((:alg_get_con_duals, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), bp.h, lam, mu))
        for (g, pr) in enumerate(bp.probs)
            each_conval(pr.game_con) do cv, i
                pos = abi_position(bp, cv, i)
                for l in eachindex(cv.inds)
                    cv.λ[l] .= [lam[pos(l, r), g] for r in eachindex(cv.λ[l])]
                    cv.μ[l] .= [mu[pos(l, r), g] for r in eachindex(cv.μ[l])]
                end
            end
            Algames.evaluate!(pr.game_con, pr.pdtraj.pr)             # conval.vals at the final iterate, as solver_methods.jl:57 leaves them
        end
    end
    # prob.stats: one record! per stored record (statistics.jl:30-57); the violation objects carry the recorded maxima (what the
    # solver's exit test and the plot recipes read, solver_plots.jl:83-125); the final record also gets its per-knot profiles
    cap = max(Int(maximum(s.records for s in stats)), 1)
    rec = Vector{AlgRecord}(undef, cap); cnt = Ref{Int32}(0)
    # the .vio vectors of the four violation objects at the final iterate (violations.jl), all games in one call
    vdyn = zeros(N - 1, B); vcon = zeros(N - 1, B); vsta = zeros(N, B); vopt = zeros(N, B)
    check(ccall((:alg_get_violation_profile, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), bp.h, vdyn, vcon, vsta, vopt))
    for (g, pr) in enumerate(bp.probs)
        check(ccall((:alg_get_history, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{AlgRecord}, Ref{Int32}), bp.h, g - 1, cap, rec, cnt))
        cnt[] < stats[g].records && @warn "AlgamesHIP: Statistics history truncated" game = g kept = cnt[] records = stats[g].records
        Algames.reset!(pr.stats)
        for t in 1:cnt[]
            r = rec[t]
            dv = Algames.DynamicsViolation(N); dv.max = r.dyn_vio
            cvv = Algames.ControlViolation(N); cvv.max = r.con_vio
            sv = Algames.StateViolation(N); sv.max = r.sta_vio
            ov = Algames.OptimalityViolation(N); ov.max = r.opt_vio
            Algames.record!(pr.stats, r.t_elap, r.res, r.delta, dv, cvv, sv, ov, Int(r.outer))      # t_elap: device real-time counter
        end
        if cnt[] > 0                                                # final record (solver_methods.jl:63): per-knot profiles from the device
            pr.stats.dyn_vio[end].vio .= view(vdyn, :, g); pr.stats.con_vio[end].vio .= view(vcon, :, g)
            pr.stats.sta_vio[end].vio .= view(vsta, :, g); pr.stats.opt_vio[end].vio .= view(vopt, :, g)
        end
    end
    return stats
end

"""
    newton_solve!(bp::BatchedGameProblem; game_id0=0, init=true)

Batched drop-in for `newton_solve!(prob)` (src/problem/solver_methods.jl:5-65).  `init=false` keeps the controls / duals the
problems currently hold as the initial guess (`alg_set_traj`); with `opts.dual_reset == false` the constraint multipliers and
penalties the problems hold are pushed first (warm start).  An `opts.f_init` other than `rand` is honoured through the reference's
own `init_traj!` on the host followed by a solve with `init = 0`.  Returns the per-game `AlgGameStats`.
"""
function newton_solve!(bp::BatchedGameProblem; game_id0::Integer=0, init::Bool=true, async::Bool=false)
    sync_options!(bp)
    B = length(bp.probs); ps = bp.probs[1].probsize
    bp.probs[1].opts.dual_reset || push_duals!(bp)
    if init && bp.probs[1].opts.f_init !== rand                     # caller-supplied generator (options.jl:11): the reference's own
        for pr in bp.probs                                          # init_traj! makes the guess (solver_methods.jl:12-13), the device
            o = pr.opts                                             # generates only the default `rand`
            Algames.Random.seed!(o.seed)
            Algames.init_traj!(pr.pdtraj; x0=pr.x0, f=o.f_init, amplitude=o.amplitude_init, s=o.shift)
        end
        init = false
    end
    if !init || bp.probs[1].opts.shift < ps.N                       # explicit initial guess / shifted warm start: upload pdtraj
        z = zeros(ps.n + ps.S, B)
        for (g, pr) in enumerate(bp.probs)
            z[1:ps.n, g] = Vector(Algames.RobotDynamics.state(pr.pdtraj.pr[1]))
            Algames.get_traj!(pr.core, view(z, ps.n+1:ps.n+ps.S, g), pr.pdtraj)
        end
        check(ccall((:alg_set_traj, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), bp.h, 0, z))
    end
    stats = Vector{AlgGameStats}(undef, B)
    if async                                                        # ShardedGameProblem: every device is launched before the first wait
        check(ccall((:alg_newton_solve_async, LIB), Cint, (Ptr{Cvoid}, Int32, Int64), bp.h, init ? 1 : 0, game_id0))
        return nothing
    end
    check(ccall((:alg_newton_solve, LIB), Cint, (Ptr{Cvoid}, Int32, Int64, Ptr{AlgGameStats}), bp.h, init ? 1 : 0, game_id0, stats))
    return pull_results!(bp, stats)
end

"Waits for an asynchronous solve of the handle and writes the results back into its problems."
function finish!(bp::BatchedGameProblem)
    check(ccall((:alg_synchronize, LIB), Cint, (Ptr{Cvoid},), bp.h))
    stats = Vector{AlgGameStats}(undef, length(bp.probs))
    check(ccall((:alg_get_stats, LIB), Cint, (Ptr{Cvoid}, Ptr{AlgGameStats}), bp.h, stats))
    return pull_results!(bp, stats)
end

"""
    ShardedGameProblem(probs; devices=0:7)

The batch split over several devices (BASELINE north_star: "the batch shards naturally across the 8 GPUs"; SURVEY.md 8(e)): shard r
owns the contiguous problems `cuts[r]` and the global scenario ids `game_id0 + first(cuts[r]) - 1 ...`, one `BatchedGameProblem`
(device handle + stream) per entry of `devices`.  `newton_solve!` launches every shard (`alg_newton_solve_async`) before it waits
for the first one; there is no data-path collective.  Python twin: `algames.jl_amd/sharding.py` (`ShardedGameProblem`,
tests/test_gpu_boundary.py::test_multi_device_solve_behind_the_boundary_two_handles_on_one_gpu).
"""
struct ShardedGameProblem{P<:GameProblem}
    shards::Vector{BatchedGameProblem{P}}
    cuts::Vector{UnitRange{Int}}
end
function ShardedGameProblem(probs::Vector{P}; devices=0:0) where {P<:GameProblem}
    B, W = length(probs), length(devices)
    per = cld(B, W)                                                 # scenarios.shard_range: ceil(B / W) games per shard, last ones shorter
    cuts = [min((r - 1) * per, B)+1:min(r * per, B) for r in 1:W]
    keep = [r for r in 1:W if !isempty(cuts[r])]
    shards = [BatchedGameProblem(probs[cuts[r]]; device=collect(devices)[r]) for r in keep]
    # one kernel shape for all shards (the automatic choice depends on a handle's batch size; the shapes differ at rounding level)
    w = Ref{Int32}(0)
    check(ccall((:alg_get_waves_per_game, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}), shards[1].h, w))
    for bp in shards
        check(ccall((:alg_set_waves_per_game, LIB), Cint, (Ptr{Cvoid}, Int32), bp.h, w[]))
    end
    ShardedGameProblem{P}(shards, cuts[keep])
end
Base.close(sp::ShardedGameProblem) = foreach(close, sp.shards)
function newton_solve!(sp::ShardedGameProblem; game_id0::Integer=0, init::Bool=true)
    for (bp, r) in zip(sp.shards, sp.cuts)
        newton_solve!(bp; game_id0=game_id0 + first(r) - 1, init=init, async=true)
    end
    return vcat([finish!(bp) for bp in sp.shards]...)
end

function newton_solve!(probs::Vector{<:GameProblem}; device::Integer=0, game_id0::Integer=0)
    bp = BatchedGameProblem(probs; device=device)
    try
        return newton_solve!(bp; game_id0=game_id0)
    finally
        close(bp)
    end
end

"ibr_newton_solve!(prob; ibr_opts) (solver_methods.jl:133-169) for every problem of the batch."
function ibr_newton_solve!(bp::BatchedGameProblem; ibr_opts::IBROptions=IBROptions(), game_id0::Integer=0)
    sync_options!(bp)
    p = bp.probs[1].probsize.p
    ordering = Int32.(ibr_opts.ordering[1:p] .- 1)
    stats = Vector{AlgGameStats}(undef, length(bp.probs))
    check(ccall((:alg_ibr_newton_solve, LIB), Cint, (Ptr{Cvoid}, Int32, Int64, Int32, Ptr{Int32}, Float64, Ptr{AlgGameStats}),
                bp.h, 1, game_id0, ibr_opts.ibr_iter, ordering, ibr_opts.Δ_min, stats))
    return pull_results!(bp, stats)
end

"ibr_newton_solve!(prob, i) (solver_methods.jl:171-228): player i best-responds on the stored trajectories."
function ibr_newton_solve!(bp::BatchedGameProblem, i::Int)
    sync_options!(bp)
    stats = Vector{AlgGameStats}(undef, length(bp.probs))
    check(ccall((:alg_ibr_solve_player, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{AlgGameStats}), bp.h, i - 1, stats))
    return pull_results!(bp, stats)
end

"""
    mpc_solve!(bp, steps; game_id0=0) -> (newton_iters, converged, states)

Receding-horizon loop of BASELINE config 5, one launch: `steps` x (newton_solve! from the shifted warm start, x0 <- RK2(x_1, u_1)),
first solve with the handle's shift / dual_reset, later ones with shift = 1 and dual_reset = false (the reference's hooks,
options.jl:16-17, primal_dual_traj.jl:29-44, solver_methods.jl:25).  `states` is n x B x (steps + 1).
"""
function mpc_solve!(bp::BatchedGameProblem, steps::Integer; game_id0::Integer=0)
    sync_options!(bp)
    B = length(bp.probs); n = bp.probs[1].probsize.n
    iters = zeros(Int64, B); conv = zeros(Int64, B)
    check(ccall((:alg_mpc_totals, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Int32), bp.h, iters, conv, 1))      # reset the totals
    states = zeros(n, B, steps + 1)
    check(ccall((:alg_mpc_solve, LIB), Cint, (Ptr{Cvoid}, Int32, Int64, Ptr{Float64}), bp.h, steps, game_id0, states))
    check(ccall((:alg_mpc_totals, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Int32), bp.h, iters, conv, 0))
    stats = Vector{AlgGameStats}(undef, B)
    check(ccall((:alg_get_stats, LIB), Cint, (Ptr{Cvoid}, Ptr{AlgGameStats}), bp.h, stats))
    pull_results!(bp, stats)                                        # the last solve's iterate, multipliers and statistics
    for (g, pr) in enumerate(bp.probs)
        pr.x0 = SVector{n}(states[:, g, end])
    end
    return iters, conv, states
end

# ---- step-wise entry points on a handle (same names as the reference's functions they batch) ------------------------------------
"residual!(prob, pdtraj) + residual_norm (global_quantities.jl:9-97): S x B residuals in vertical order and ||res||_1 / S per game."
function residual(bp::BatchedGameProblem; reg::Float64=0.0)
    S = bp.probs[1].probsize.S; B = length(bp.probs)
    res = zeros(S, B); rn = zeros(B)
    check(ccall((:alg_residual, LIB), Cint, (Ptr{Cvoid}, Int32, Float64, Ptr{Float64}, Ptr{Float64}), bp.h, 0, reg, res, rn))
    return res, rn
end
"residual_jacobian! + regularisation (global_quantities.jl:109-193): S x S x B dense, rows vertical order, columns horizontal order."
function residual_jacobian(bp::BatchedGameProblem; reg::Float64=0.0, games::UnitRange{Int}=1:length(bp.probs))
    S = bp.probs[1].probsize.S
    jac = zeros(S, S, length(games))      # only the requested games are built on the device and copied
    check(ccall((:alg_residual_jacobian_games, LIB), Cint, (Ptr{Cvoid}, Float64, Int32, Int32, Ptr{Float64}),
                bp.h, reg, Int32(first(games) - 1), Int32(length(games)), jac))
    return jac
end
"""
    set_refinement!(bp; max_steps = 2, tol = 2.0^-34, mu_tight = 1.6e5)

Iterative refinement of the Newton direction (alg_set_refinement): the stand-in for the backward stability of `lu(core.jac) \\ core.res`
(solver_methods.jl:87).  After every structured solve the opt-u rows of `J d = -res` are evaluated; while their row-wise backward error exceeds `tol`
the direction is corrected by one more elimination on the residual (at most `max_steps` times); below `mu_tight` (largest penalty of the game)
the tolerance is relaxed in proportion, at most 256 x.  `max_steps = 0` switches both off.
"""
set_refinement!(bp::BatchedGameProblem; max_steps::Integer=2, tol::Float64=2.0^-34, mu_tight::Float64=1.6e5) =
    check(ccall((:alg_set_refinement, LIB), Cint, (Ptr{Cvoid}, Int32, Float64, Float64), bp.h, max_steps, tol, mu_tight))

"""
    set_handoff!(bp, iters)

Straggler hand-off for heterogeneous batches (alg_set_handoff): games that need more than `iters` inner iterations in the one-wavefront kernel
park and a second launch finishes them with the team kernel; `0` switches it off (the default).  `handoff(bp)` returns the budget and the
number of games the most recent solve handed over.
"""
set_handoff!(bp::BatchedGameProblem, iters::Integer) = check(ccall((:alg_set_handoff, LIB), Cint, (Ptr{Cvoid}, Int32), bp.h, iters))
function handoff(bp::BatchedGameProblem)
    k = Ref{Int32}(0); n = Ref{Int32}(0)
    check(ccall((:alg_get_handoff, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), bp.h, k, n))
    return Int(k[]), Int(n[])
end
"Line search with the step sizes tried in groups (`true`, default) or one after another (`false`): bit-identical norms (alg_set_line_search_groups)."
set_line_search_groups!(bp::BatchedGameProblem, on::Bool) = check(ccall((:alg_set_line_search_groups, LIB), Cint, (Ptr{Cvoid}, Int32), bp.h, on ? 1 : 0))

"3 x B: [max |rho|, row-wise backward error, largest row scale] of the opt-u rows of every game's last Newton direction (alg_get_direction_gate)."
function direction_gate(bp::BatchedGameProblem)
    out = zeros(3, length(bp.probs))
    check(ccall((:alg_get_direction_gate, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), bp.h, out))
    return out
end

"Give the inspection entry points' device scratch (dense Jacobians, MPC state logs) back to the allocator."
release_scratch!(bp::BatchedGameProblem) = check(ccall((:alg_release_scratch, LIB), Cint, (Ptr{Cvoid},), bp.h))

end # module
