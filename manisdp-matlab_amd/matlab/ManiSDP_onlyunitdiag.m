% Drop-in for src/primal/ManiSDP_onlyunitdiag.m of wangjie212/ManiSDP-matlab:
%   Min <C, X>  s.t.  X >= 0, diag(X) = 1
% Same signature, option names, defaults, printed lines and data fields.  The outer loop
% and the bookkeeping are the reference's; every trustregions(problem, Y, opts) call is
% replaced by the device-resident RTR/tCG solve of libmanisdp_hip (manisdp_mex('rtr', ...)),
% and for n > options.dense_eig_max the dense eig(full(S)) by the few-eigenvector escape.
function [X, obj, data] = ManiSDP_onlyunitdiag(C, options)

if ~isfield(options,'p0'); options.p0 = 2; end
if ~isfield(options,'AL_maxiter'); options.AL_maxiter = 20; end
if ~isfield(options,'tol'); options.tol = 1e-8; end
if ~isfield(options,'theta'); options.theta = 1e-1; end
if ~isfield(options,'delta'); options.delta = 8; end
if ~isfield(options,'alpha'); options.alpha = 0.5; end
if ~isfield(options,'tolgradnorm'); options.tolgradnorm = 1e-8; end
if ~isfield(options,'TR_maxinner'); options.TR_maxinner = 100; end
if ~isfield(options,'TR_maxiter'); options.TR_maxiter = 40; end
if ~isfield(options,'line_search'); options.line_search = 0; end
if ~isfield(options,'dense_eig_max'); options.dense_eig_max = 3000; end

fprintf('ManiSDP is starting...\n');
n = size(C,1);
fprintf('SDP size: n = %i, m = %i\n', n, n);

h = manisdp_mex('create_onlyunitdiag', C);
cleanup = onCleanup(@() manisdp_mex('destroy', h));
p = options.p0;
Y = randn(p, n);  Y = Y./sqrt(sum(Y.^2, 1));      % M.rand() of obliquefactoryNTrans
U = [];
opts.maxinner = options.TR_maxinner;
opts.maxiter = options.TR_maxiter;
opts.tolgradnorm = options.tolgradnorm;

data.status = 0;
timespend = tic;
for iter = 1:options.AL_maxiter
    if ~isempty(U)
        Y = line_search(Y, U);
    end
    [Y, info] = manisdp_mex('rtr', h, Y, opts);
    gradnorm = info.gradnorm;
    z = manisdp_mex('get_z', h);                     % z = sum((Y*C).*Y)
    obj = full(sum(z));
    if n <= options.dense_eig_max
        S = C - diag(z);
        [vS, dS] = eig(full(S), 'vector');
    else
        [dS, vS, lmax] = manisdp_mex('escape_eigs', h, options.delta, 1e-9, 60000);
        dS = [dS; lmax];                             % dS(1) = lambda_min, dS(end) = lambda_max
        S = [];
    end
    dinf = max(0, -dS(1))/(1+dS(end));
    e = sqrt(max(eig(Y*Y'), 0)); e = sort(e, 'descend');      % singular values via the p x p Gram matrix
    [Qg, Dg] = eig(Y*Y'); [~, order] = sort(diag(Dg), 'descend'); Qg = Qg(:, order);
    r = sum(e >= options.theta*e(1));
    fprintf('Iter %d, obj:%0.8f, dinf:%0.1e, r:%d, p:%d, time:%0.2fs\n', ...
             iter,    obj,       dinf,       r,    p,    toc(timespend));
    if dinf < options.tol
        fprintf('Optimality is reached!\n');
        break;
    end
    if mod(iter, 20) == 0
        if iter > 50 && dinf > dinf0
            data.status = 2;
            fprintf('Slow progress!\n');
            break;
        else
            dinf0 = dinf;
        end
    end
    if r <= p - 1
        Y = Qg(:,1:r)'*Y;                            % = V(:,1:r)'.*e(1:r) of the reference
        p = r;
    end
    nne = max(min(sum(dS(1:min(end-1,options.delta)) < 0), options.delta), 1);
    if options.line_search == 1
        U = [zeros(p, n); vS(:,1:nne)'];
    end
    p = p + nne;
    if options.line_search == 1
        Y = [Y; zeros(nne,n)];
    else
        Y = [Y; options.alpha*vS(:,1:nne)'];
        Y = Y./sqrt(sum(Y.^2));
    end
end
X = Y'*Y;
data.X = X;
data.S = S;
data.z = z;
data.dinf = dinf;
data.gradnorm = gradnorm;
data.time = toc(timespend);
if data.status == 0 && dinf > options.tol
    data.status = 1;
    fprintf('Iteration maximum is reached!\n');
end
fprintf('ManiSDP: optimum = %0.8f, time = %0.2fs\n', obj, toc(timespend));

    function nY = line_search(Y, U)
         alpha = 1;
         cost0 = manisdp_mex('linesearch_cost', h, Y, U, 0);
         i = 1;
         nY = Y + alpha*U;  nY = nY./sqrt(sum(nY.^2));
         while i <= 15 && manisdp_mex('linesearch_cost', h, Y, U, alpha) - cost0 > -1e-3
              alpha = 0.8*alpha;
              nY = Y + alpha*U;  nY = nY./sqrt(sum(nY.^2));
              i = i + 1;
         end
    end
end
